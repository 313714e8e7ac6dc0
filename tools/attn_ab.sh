#!/bin/bash
# Developer tool (build container): variant libraries of the attention kernels for an interleaved A/B on ONE box.
#   tools/attn_ab.sh name1 "-DFLAG=.." name2 "-DFLAG=.." ...   ->  mobgt_amd/libmobgt_hip_ab_<name>.so each
# then on the GPU box: python tools/attn_ab.py name1 name2 ...  (the shipped library is always variant "ship")
set -e
cd "$(dirname "$0")/../mobgt_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -I../../include --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
OBJS=$(ls *.o | grep -v '^attn.o$')
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc $FLAGS $2 -c attn.hip -o /tmp/attn_ab_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmobgt_hip_ab_$1.so /tmp/attn_ab_$1.o $OBJS
  echo built libmobgt_hip_ab_$1.so "($2)"
  shift 2
done
