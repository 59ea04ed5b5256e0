#!/bin/bash
# Builds a variant of the render library with extra compiler flags for the wave kernel and the device half of the ABI:
#   tools/build_variant.sh NAME [-DFLAG ...]   ->  build/variants/librt_amd_NAME.so   (select it with RT_AMD_LIB)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=raytracer_amd/csrc
mkdir -p build/variants
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Iinclude "$@" \
  -c ${WAVE_SRC:-$C/rt_kernel_wave.hip} -o build/variants/wave_$name.o
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Iinclude "$@" \
  -c $C/rt_device.hip -o build/variants/device_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/librt_amd_$name.so \
  $C/rt_host.o $C/rt_image.o build/variants/device_$name.o $C/rt_kernel_lane.o $C/rt_queue.o $C/rt_gather.o build/variants/wave_$name.o -lz -ldl -pthread
rm -f build/variants/wave_$name.o build/variants/device_$name.o
echo build/variants/librt_amd_$name.so
