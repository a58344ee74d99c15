# Development aid: runs tools/ubench/fused_stamps.py over the stamped library variants named on the command line
# (tools/ubench/libvts_fstamp_<tag>.so, built by build_fstamp.sh):  bash tools/ubench/run_fused_variants.sh 256 st nt
set -e
mkdir -p gpurun_out
cin=$1; shift
for v in "$@"; do
  VTS_STAMP_LIB=$PWD/tools/ubench/libvts_fstamp_$v.so timeout -k 10 120 python tools/ubench/fused_stamps.py $cin > gpurun_out/r3b_stamp_${v}_$cin.txt 2>&1
  echo "== $v"; cat gpurun_out/r3b_stamp_${v}_$cin.txt
done
