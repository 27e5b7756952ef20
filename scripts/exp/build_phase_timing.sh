#!/bin/bash
# experiment build: libmadnet_hip with -DMH_PHASE_TIMING (per-workgroup s_memtime stamps inside conv_igemm_kernel) -> scripts/exp/libmadnet_hip_phase.so
set -e
cd "$(dirname "$0")/../../real-time-self-adaptive-deep-stereo_amd/csrc"
OUT=../../scripts/exp
mkdir -p /tmp/mh_phase
for f in lib conv conv_patch conv_direct wgrad corr shift_corr ops; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -DMH_PHASE_TIMING -c $f.hip -o /tmp/mh_phase/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmadnet_hip_phase.so /tmp/mh_phase/*.o
ls -la $OUT/libmadnet_hip_phase.so
