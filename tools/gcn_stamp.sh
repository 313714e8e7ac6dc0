#!/bin/bash
# Developer tool: diagnostic build of the small-GCN kernels with in-kernel stamps (-DSG_DEBUG) as a SEPARATE library
# (mobgt_amd/libmobgt_hip_sgstamp.so); on the GPU box: python tools/gcn_stamp.py
set -e
cd "$(dirname "$0")/../mobgt_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -I../../include --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS -DSG_DEBUG -DSG_WG=${SG_WG:-0} -c smallgcn.hip -o /tmp/sg_stamp.o
OBJS=$(ls *.o | grep -v '^smallgcn.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmobgt_hip_sgstamp.so /tmp/sg_stamp.o $OBJS
echo built ../libmobgt_hip_sgstamp.so
