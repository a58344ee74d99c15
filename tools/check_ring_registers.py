#!/usr/bin/env python3
"""Static check of the bf16x3 recurrent kernels' W_hh register ring on the generated gfx950 ISA.

The ring buffers are filled by inline-asm global loads whose data arrives asynchronously; the
kernel is only correct if, between a load and the hand-counted s_waitcnt that proves it landed,
nothing reads or copies those registers.  This script compiles the source to assembly and checks
for every recurrent kernel with such loads that no instruction (MFMA, move, spill, another load) touches a
register written by such a load (`global_load_dwordx4 v[a:b] or a[a:b], vN, s[c:d] offset:K`, and the pair
kernel's `buffer_load_dwordx4 v[a:b], vN, s[c:f], 0 offen ...` of its peer's partial sums) while the
load can still be outstanding, which also re-derives every hand-counted vmcnt from the
instruction stream.  Exit status 1 and a listing on any violation.

    python tools/check_ring_registers.py [kernel sources ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def kernels(asm):
    cur, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w*lstm_rec\w*_kernel\w*):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(line)
            if "s_endpgm" in line:
                yield cur, body
                cur = None


def check(name, body):
    """Forward dataflow over the kernel's control-flow graph.  The state is the FIFO of
    outstanding vector-memory loads, oldest first: ring loads with the registers they write, every other
    load (global / buffer / scratch / flat) as an entry without registers.  Loads return in order, so an
    s_waitcnt vmcnt(N) leaves at most the N youngest of them outstanding.  Stores are left out: they share the
    counter but may retire early, so counting on them would be unsound, while ignoring them only makes a
    given vmcnt(N) stricter than the model assumes.  A register is in flight from its ring load until such a
    wait retires it; any instruction touching an in-flight register is an error."""
    code, labels, from_asm, in_asm = [], {}, set(), False
    for l in body:
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") and not l.endswith(":"):
            continue
        if l.endswith(":"):
            labels[l[:-1]] = len(code)
            continue
        if in_asm:
            from_asm.add(len(code))       # instruction text written by the kernel source, not the compiler
        code.append(l)
    ring_re = re.compile(r"^(?:global_load_dwordx4 ([va]\[\d+:\d+\]), v\d+, s\[\d+:\d+\]"
                         r"|buffer_load_dwordx4 ([va]\[\d+:\d+\]), v\d+, s\[\d+:\d+\], 0 offen)")
    is_ring = [bool(ring_re.match(l)) and i in from_asm for i, l in enumerate(code)]
    other_load_re = re.compile(r"^(global|buffer|scratch|flat)_load")
    if not any(is_ring):
        return None                       # a kernel without an asm ring: nothing to check

    # basic blocks
    br = re.compile(r"^s_(cbranch_\w+|branch) (\S+)$")
    leaders = {0} | set(labels.values())
    for i, l in enumerate(code):
        if br.match(l) or l.startswith("s_endpgm"):
            leaders.add(i + 1)
    starts = sorted(x for x in leaders if x < len(code))
    block_of = {s: (s, e) for s, e in zip(starts, starts[1:] + [len(code)])}

    def succs(s):
        e = block_of[s][1]
        last = code[e - 1]
        m = br.match(last)
        out = []
        if last.startswith("s_endpgm"):
            return out
        if m:
            if m.group(2) in labels:
                out.append(labels[m.group(2)])
            if m.group(1) != "branch" and e < len(code):
                out.append(e)
        elif e < len(code):
            out.append(e)
        return out

    bad, seen = [], set()

    def run_block(s, fifo, report):
        fifo = [set(x) for x in fifo]
        for i in range(*block_of[s]):
            l = code[i]
            op = l.split()[0]
            inflight = set().union(*fifo) if fifo else set()
            if is_ring[i]:
                mm = ring_re.match(l)
                dst = regs_of(mm.group(1) or mm.group(2))
                if dst & inflight and report and l not in seen:
                    seen.add(l)
                    bad.append(f"{name}: `{l}` reloads registers still in flight")
                fifo.append(dst)
            elif other_load_re.match(l):
                hit = regs_of(l[len(op):]) & inflight
                if hit and report and l not in seen:
                    seen.add(l)
                    bad.append(f"{name}: `{l}` touches in-flight ring registers {sorted(hit)[:8]}")
                fifo.append(set())
            elif op == "s_waitcnt":
                w = re.search(r"vmcnt\((\d+)\)", l)
                if w:
                    del fifo[:max(0, len(fifo) - int(w.group(1)))]
            elif not op.startswith("s_"):
                hit = regs_of(l[len(op):]) & inflight
                if hit and report and l not in seen:
                    seen.add(l)
                    bad.append(f"{name}: `{l}` touches in-flight ring registers {sorted(hit)[:8]}")
        return tuple(frozenset(x) for x in fifo)

    def merge(a, b):
        if a is None or a == b:
            return b if a is None else a
        # differing histories: keep the longer queue, position-wise union from the young end
        if len(a) < len(b):
            a, b = b, a
        pad = len(a) - len(b)
        return tuple(a[i] | (b[i - pad] if i >= pad else frozenset()) for i in range(len(a)))

    state_in = {s: None for s in starts}
    state_in[0] = ()
    work = [0]
    rounds = 0
    while work and rounds < 20000:
        rounds += 1
        s = work.pop()
        out = run_block(s, state_in[s], False)
        for t in succs(s):
            m = merge(state_in[t], out)
            if m != state_in[t]:
                state_in[t] = m
                work.append(t)
    if work:
        bad.append(f"{name}: dataflow did not converge")
    for s in starts:
        if state_in[s] is not None:
            run_block(s, state_in[s], True)
    return bad


def main():
    csrc = os.path.join(ROOT, "vae-time-series_amd", "csrc")
    srcs = sys.argv[1:] or [os.path.join(csrc, "lstm_rec_bf16x3.hip"), os.path.join(csrc, "lstm_rec_f32.hip")]
    problems, n = [], 0
    for src in srcs:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out,
                            src], check=True, stderr=subprocess.DEVNULL)
            asm = open(out).read()
        for name, body in kernels(asm):
            res = check(name, body)
            if res is None:
                continue
            n += 1
            problems += res
    if n == 0:
        problems.append("no kernel with an asm register ring found in the assembly")
    for p in problems:
        print(p)
    print(f"checked {n} kernels: {'FAILED' if problems else 'ok'}")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
