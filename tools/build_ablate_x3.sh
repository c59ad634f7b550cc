#!/bin/bash
# Timing ablations of k_pconv_x3 (INVALID results): -DMPF_X3_ABLATE=1 no MFMA, 2 no activation split, 3 no per-step barrier.  A/B on one box:
#   bash tools/build_ablate_x3.sh && for v in 1 2 3; do MPIFLOW_HIP_LIB=$PWD/mpiflow_amd/libmpiflow_ablate_x3_$v.so python tools/bench_precise.py x3-notile; done
set -e
cd "$(dirname "$0")/../mpiflow_amd/csrc"
make -j4 > /dev/null
for v in 1 2 3; do
  /opt/rocm/bin/hipcc -DMPF_X3_ABLATE=$v -I../../include -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
      -fno-slp-vectorize -c mpf_pconv.hip -o /tmp/mpf_pconv_ablate_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmpiflow_ablate_x3_$v.so mpf_render.o mpf_generic.o mpf_fwarp.o mpf_conv.o mpf_encoder.o /tmp/mpf_pconv_ablate_$v.o \
      mpf_frames.o mpf_inpaint.o
done
ls ../libmpiflow_ablate_x3_*.so
