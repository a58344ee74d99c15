"""Development aid: run bench.py against another build of the library (A/B of kernel variants in one GPU call).

    python tools/ubench/bench_with_lib.py path/to/libvts_variant.so [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
lib = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
from vts import _lib as L
L.LIB_PATH = lib
import bench
bench.main()
