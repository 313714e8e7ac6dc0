#!/bin/bash
# Developer tool: diagnostic build of the attention forward with in-kernel cycle stamps (-DATTN_STAMP) as a SEPARATE library
# (mobgt_amd/libmobgt_hip_stamp.so; the shipped library never contains stamps), then -- on the GPU box -- tools/attn_stamp.py.
set -e
cd "$(dirname "$0")/../mobgt_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -I../../include --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS -DATTN_STAMP $STAMP_EXTRA -c attn.hip -o /tmp/attn_stamp.o
OBJS=$(ls *.o | grep -v '^attn.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmobgt_hip_stamp.so /tmp/attn_stamp.o $OBJS
echo built ../libmobgt_hip_stamp.so
