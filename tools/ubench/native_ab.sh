# Development aid: recorded native-shape step times with the in-tree library and with variant builds (on the GPU box).
#   bash tools/ubench/native_ab.sh [lib.so ...]
set -e
cd $GRAFT_REPO_ROOT
echo "in-tree:"; timeout -k 10 200 python tools/ubench/native_graph_time.py 2>&1 | grep "graph=True"
for lib in "$@"; do
  echo "$lib:"; VTS_LIB=$lib timeout -k 10 200 python tools/ubench/native_graph_time.py 2>&1 | grep "graph=True"
done
