#!/bin/bash
# Extra builds of the library whose convolution engine has one part removed at compile time (-DMPF_CONV_ABLATE=<bits>, see mpf_conv.hip: 2 no MFMAs / 4 loader
# arithmetic replaced by a copy / 8 no barriers in the chunk loop / 16 no global -> LDS copies; INVALID results).  A/B on one box:
#   bash tools/build_ablate_conv.sh 2 4 8 16 && for b in 2 4 8 16; do MPIFLOW_HIP_LIB=$PWD/mpiflow_amd/libmpiflow_ablate_conv$b.so python tools/ab_engine_pw.py ""; done
set -e
cd "$(dirname "$0")/../mpiflow_amd/csrc"
make -j4 > /dev/null 2>&1
for b in "$@"; do
  ( /opt/rocm/bin/hipcc -DMPF_CONV_ABLATE=$b -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
      -fno-slp-vectorize -c mpf_conv.hip -o /tmp/mpf_conv_ablate$b.o 2> /dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmpiflow_ablate_conv$b.so mpf_render.o mpf_generic.o mpf_fwarp.o /tmp/mpf_conv_ablate$b.o mpf_encoder.o mpf_pconv.o \
      mpf_frames.o mpf_inpaint.o
    echo ../libmpiflow_ablate_conv$b.so ) &
done
wait
