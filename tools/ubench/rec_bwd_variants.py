"""Development aid: where does the one-workgroup backward recurrence spend its step?  Builds copies of the library whose
backward kernel has one part taken out (results are garbage, addresses and launch shape unchanged) for timing with
tools/bench_kernels.py --mode 1 --only recurrent_bwd --lib tools/ubench/librec_xN.so.

    1  no W_hh stream (ring loads and waits removed)
    2  cell backward without arithmetic (the LDS writes and splits stay)
    3  no MFMAs
    4  no loads of the saved gates / cells / dh
    5  1 + 4: only MFMAs, cell arithmetic, LDS traffic, barriers
    s  the library as it is, with the staged kernel's phase stamps compiled in (-DVTS_RB_STAMP; read them with
       tools/ubench/rec_bwd_stamps.py)
(1-5 patch lstm_rec3_bwd_kernel, the kernel VTS_LSTM_STAGED_BWD=0 selects.)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "vae-time-series_amd", "csrc")
src = open(os.path.join(CSRC, "lstm_rec_bf16x3.hip")).read()
a = src.index("// ---- backward ---")
b = src.index("template <int H, bool PL>", src.index("lstm_rec3_bwd_kernel"))
head, body, tail = src[:a], src[a:b], src[b:]


def variant(n):
    s = body
    if n in (1, 5):
        s = re.sub(r"w_issue<UT, RING_AG>\([^;]*;", ";", s)
        s = re.sub(r"w_wait<[^;]*;", ";", s)
    if n == 2:
        s = s.replace("const float tc = fast_tanh(ct[reg]);", "const float tc = ct[reg];")
        s = re.sub(r"dai\[reg\] = .*;", "dai[reg] = d_i;", s)
        s = re.sub(r"daf\[reg\] = .*;", "daf[reg] = d_f;", s)
        s = re.sub(r"dag\[reg\] = .*;", "dag[reg] = d_g;", s)
        s = re.sub(r"dao\[reg\] = .*;", "dao[reg] = d_o;", s)
        s = s.replace("const float dcc = dc[jl][reg] + dht[reg] * go[reg] * (1.f - tc * tc);", "const float dcc = dc[jl][reg] + dht[reg];")
    if n == 3:
        s = re.sub(r"acc\[jl\] = __builtin_amdgcn_mfma[^;]*;", "acc[jl][0] += __builtin_bit_cast(float, w[jl][0] ^ ah[0] ^ al[1] ^ w[JPW + jl][1]);", s)
    if n in (4, 5):
        s = s.replace("gv[jl * 4 + q] = ld_stream(Gt + (long)(q * C::JT + jt) * 256);", "gv[jl * 4 + q] = f32x4{0.5f, 0.5f, 0.5f, 0.5f};")
        s = s.replace("dv[jl] = ld_stream(dhseq + ((long)(bt * T + t) * C::JT + jt) * 256 + lane * 4);", "dv[jl] = f32x4{0.f, 0.f, 0.f, 0.f};")
        s = s.replace("cv[jl] = ld_stream(cseq + ((long)(bt * T + t) * C::JT + wave * JPW + jl) * 256 + lane * 4);", "cv[jl] = f32x4{0.1f, 0.1f, 0.1f, 0.1f};")
    assert s != body
    return head + s + tail


objs = "vts_api.o gemm_f32.o lstm_rec_f32.o lstm_rec_cluster.o lstm_planes.o mlp_chain.o elementwise.o attention.o ffn_fused.o windows.o mixture3.o".split()
os.makedirs("/tmp/recx", exist_ok=True)
for n in (sys.argv[1:] or "1 2 3 4 5".split()):
    p = os.path.join(CSRC, "_recx%s.hip" % n)
    open(p, "w").write(src if n == "s" else variant(int(n)))
    o = "/tmp/recx/recx%s.o" % n
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-c", p, "-o", o]
                          + (["-DVTS_RB_STAMP"] if n == "s" else []), cwd=CSRC)
    os.remove(p)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                           os.path.join(ROOT, "tools", "ubench", "librec_x%s.so" % n), o] + objs, cwd=CSRC)
    print("built", n)
