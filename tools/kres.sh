#!/bin/bash
# usage: tools/kres.sh file.hip  -> per-kernel VGPR/AGPR/scratch/LDS/occupancy table
f=$1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I/root/repo/include -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(line)print line; line=$3; next}{line=line" | "$0}END{print line}'
