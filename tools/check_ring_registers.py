#!/usr/bin/env python3
"""Static check of the bf16x3 recurrent kernels' W_hh register ring on the generated gfx950 ISA.

The ring buffers are filled by inline-asm global loads whose data arrives asynchronously; the
kernel is only correct if, between a load and the hand-counted s_waitcnt that proves it landed,
nothing reads or copies those registers.  This script compiles the source to assembly and checks
for every lstm_rec3_* kernel that no instruction (MFMA, move, spill, another load) touches a
register written by such a load (`global_load_dwordx4 v[a:b], vN, s[c:d] offset:K`) while the
load can still be outstanding, which also re-derives every hand-counted vmcnt from the
instruction stream.  Exit status 1 and a listing on any violation.

    python tools/check_ring_registers.py [path/to/lstm_rec_bf16x3.hip]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels(asm):
    cur, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w*lstm_rec3_\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(line)
            if "s_endpgm" in line:
                yield cur, body
                cur = None


def check(name, body):
    """Walk the kernel in program order, every loop body twice (so that state carried over a
    back-edge is seen), with a FIFO of outstanding ring loads.  Only ring loads are modelled:
    other VMEM operations can only make a given vmcnt(N) stricter than the model assumes (loads
    return in order) or, for stores, may retire early, so leaving them out is the conservative
    reading.  A register is in flight from its ring load until an s_waitcnt vmcnt(N) leaves at most
    N younger ring loads outstanding; any instruction touching an in-flight register is an error."""
    code, labels = [], {}
    for l in body:
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") and not l.endswith(":"):
            continue
        if l.endswith(":"):
            labels[l[:-1]] = len(code)
            continue
        code.append(l)
    ring_load = re.compile(r"^global_load_dwordx4 (v\[\d+:\d+\]), v\d+, s\[\d+:\d+\]")
    if not any(ring_load.match(l) for l in code):
        return [f"{name}: no ring loads found"]
    fifo, inflight, bad, seen = [], set(), [], set()

    def step(l):
        m = ring_load.match(l)
        op = l.split()[0]
        if m:
            dst = regs_of(m.group(1))
            if dst & inflight:
                bad.append(f"{name}: `{l}` reloads registers still in flight")
            fifo.append(dst)
            inflight.update(dst)
            return
        if op == "s_waitcnt":
            w = re.search(r"vmcnt\((\d+)\)", l)
            if w:
                while len(fifo) > int(w.group(1)):
                    inflight.difference_update(fifo.pop(0))
            return
        if op.startswith("s_"):
            return
        hit = regs_of(l[len(op):]) & inflight
        if hit and l not in seen:
            seen.add(l)
            bad.append(f"{name}: `{l}` touches in-flight ring registers {sorted(hit)[:8]}")

    def walk(lo, hi, depth):
        i = lo
        while i < hi:
            l = code[i]
            step(l)
            m = re.match(r"^s_cbranch_\w+ (\S+)$", l) or re.match(r"^s_branch (\S+)$", l)
            if m and m.group(1) in labels and lo <= labels[m.group(1)] <= i and depth < 4:
                walk(labels[m.group(1)], i + 1 if False else i, depth + 1)   # one more trip round the loop
            i += 1

    walk(0, len(code), 0)
    if fifo and not any("s_endpgm" in l for l in code[-3:]):
        bad.append(f"{name}: kernel text does not end in s_endpgm")
    return bad


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "vae-time-series_amd", "csrc", "lstm_rec_bf16x3.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, src],
                       check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    problems, n = [], 0
    for name, body in kernels(asm):
        n += 1
        problems += check(name, body)
    if n == 0:
        problems.append("no lstm_rec3_* kernel found in the assembly")
    for p in problems:
        print(p)
    print(f"checked {n} kernels: {'FAILED' if problems else 'ok'}")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
