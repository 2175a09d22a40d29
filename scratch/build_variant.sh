#!/bin/bash
# scratch build of the library: scratch/build_variant.sh NAME [-DKNOB=...]...  ->  scratch/lib_NAME.so
# (SRC=<dir> builds another checkout's csrc, e.g. a `git worktree` of an older commit)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
SRC=${SRC:-$ROOT/vireo_amd/csrc}
cd "$SRC"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$SRC/../../include" "$@" \
    vrx_engine.hip vrx_comm.hip vrx_host.cpp -o "$ROOT/scratch/lib_$NAME.so" -ldl -lpthread -lz
echo "built scratch/lib_$NAME.so"
