#!/bin/bash
# Developer tool: diagnostic build of the chain kernels with in-kernel stamps (-DCH_DEBUG) as a SEPARATE library
# (mobgt_amd/libmobgt_hip_chstamp.so; the shipped library never contains stamps); on the GPU box: python tools/chain_stamp.py
set -e
cd "$(dirname "$0")/../mobgt_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -I../../include --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS -DCH_DEBUG $STAMP_EXTRA -c chain.hip -o /tmp/chain_stamp.o
OBJS=$(ls *.o | grep -v '^chain.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmobgt_hip_chstamp.so /tmp/chain_stamp.o $OBJS
echo built ../libmobgt_hip_chstamp.so
