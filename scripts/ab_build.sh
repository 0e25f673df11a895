#!/bin/bash
# Build an A/B variant of the library: scripts/ab_build.sh <name> <extra hipcc flags for mmf_scan_bf16.hip...>
# -> multimodal-fusion_amd/libmmf_hg_<name>.so (select it with MMF_HG_LIBRARY=<path>)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python multimodal-fusion_amd/csrc/build.py > /dev/null
C=multimodal-fusion_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -fno-gpu-rdc -fno-honor-nans "$@" -c $C/mmf_scan_bf16.hip -o $C/_obj/mmf_scan_bf16_$name.o
objs=$(ls $C/_obj/mmf_*.o | grep -v "mmf_scan_bf16")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $objs $C/_obj/mmf_scan_bf16_$name.o -o multimodal-fusion_amd/libmmf_hg_$name.so
echo multimodal-fusion_amd/libmmf_hg_$name.so
