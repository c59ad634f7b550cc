#!/bin/bash
# A second build of the library whose convolution engine re-uses the first B fragment of a chunk for every k-step (-DMPF_CONV_ABLATE_LDS: a fifth of the
# B-fragment LDS reads, INVALID results): the upper bound of any formulation that feeds several MFMAs from one LDS read.  A/B on one box:
#   bash tools/build_ablate_lds.sh && python tools/bench_engine.py && MPIFLOW_HIP_LIB=$PWD/mpiflow_amd/libmpiflow_ablate_lds.so python tools/bench_engine.py
set -e
cd "$(dirname "$0")/../mpiflow_amd/csrc"
make -j4 > /dev/null
/opt/rocm/bin/hipcc -DMPF_CONV_ABLATE_LDS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -fno-slp-vectorize -c mpf_conv.hip -o /tmp/mpf_conv_ablate.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmpiflow_ablate_lds.so mpf_render.o mpf_generic.o mpf_fwarp.o /tmp/mpf_conv_ablate.o mpf_encoder.o mpf_pconv.o \
    mpf_frames.o mpf_inpaint.o
echo ../libmpiflow_ablate_lds.so
