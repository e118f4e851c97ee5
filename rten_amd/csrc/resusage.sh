#!/bin/bash
# Prints one compact line per kernel: name VGPRs scratch occupancy sgpr-spill vgpr-spill LDS
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
 grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|VGPRs Spill|LDS Size" | sed 's/.*remark: //; s/ \[-Rpass.*//' | \
 awk '/Function Name/{if(l)print l; l=$3} /VGPRs:/{l=l" v="$2} /ScratchSize/{l=l" scr="$4} /Occupancy/{l=l" occ="$3} /VGPRs Spill/{l=l" vsp="$3} /LDS/{l=l" lds="$4} END{print l}' | c++filt | sed 's/(anonymous namespace):://g; s/(.*GemmArgs)//'
