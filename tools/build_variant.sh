#!/bin/bash
# Builds a variant of the render library with extra compiler flags for the wave kernel and the device half of the ABI:
#   tools/build_variant.sh NAME [-DFLAG ...]   ->  build/variants/librt_amd_NAME.so   (select it with RT_AMD_LIB)
# With RT_GEN_* set in the environment (tools/gen_trace_asm.py: RT_GEN_COUNT=1, RT_GEN_BATCH=0, RT_GEN_BATCH_MIN=n, ...) the
# variant gets an assembly block of its own, generated beside a copy of the kernel source under build/variants/src_NAME.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=raytracer_amd/csrc
mkdir -p build/variants
src=${WAVE_SRC:-$C/rt_kernel_wave.hip}
if env | grep -q '^RT_GEN_'; then
  mkdir -p build/variants/src_$name
  cp $C/rt_kernel_wave.hip build/variants/src_$name/
  python3 tools/gen_trace_asm.py build/variants/src_$name/rt_trace_asm.inc > /dev/null
  src=build/variants/src_$name/rt_kernel_wave.hip
fi
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Iinclude -I$C "$@" \
  -c $src -o build/variants/wave_$name.o
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Iinclude "$@" \
  -c $C/rt_device.hip -o build/variants/device_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/librt_amd_$name.so \
  $C/rt_host.o $C/rt_image.o build/variants/device_$name.o $C/rt_kernel_lane.o $C/rt_queue.o $C/rt_gather.o build/variants/wave_$name.o -lz -ldl -pthread
rm -f build/variants/wave_$name.o build/variants/device_$name.o
echo build/variants/librt_amd_$name.so
