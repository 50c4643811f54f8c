#!/bin/bash
# usage: tools/kres.sh file.hip   -> per-kernel VGPR / AGPR / scratch / LDS / occupancy table (CPU only: hipcc cross-compiles)
#        tools/kres.sh --check    -> the instantiations the BASELINE configs dispatch to by default must have ScratchSize == 0
#                                    (VERDICT round 3, item 9); exit code 1 otherwise
C=/root/repo/online-neural-cdes_amd/csrc
table() {
  # (the one unit the Makefile builds with MFMA results in VGPRs gets the same flag here)
  case $(basename $1) in ncde_fast_fwd3.hip) X="-mllvm -amdgpu-mfma-vgpr-form";; *) X="";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $X -I/root/repo/include -c $1 -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
   | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
   | awk '/Function Name/{if(line)print line; line=$3; next}{line=line" | "$0}END{print line}'
}
if [ "$1" != "--check" ]; then table $1; exit 0; fi
# mangled-name fragments: template arguments as ILi<v>E...
#   cfg2 / cfg3: ncde_fwd_fast_bf3<32,32,20,NW4,linear,rk4,PROF0,NLT3,fp16x2,PLAN0>, ncde_adj_fast3<NL3,C20,linear,rk4,PROF0,DISC0,HP2,PLAN0>
#   cfg4:        ncde_fwd_fast_bf3<64,64,4,NW4,cubic,midpoint,0,3,fp16x2,0>,       ncde_adj_h64<midpoint,NS1,HPF1,DISC0,PLAN0>
#   dopri5:      ncde_dpf_fwd<32,32,20>, ncde_dpf_tape<32,32,20,3>   (ncde_dpf_adj keeps 128 B: DESIGN.md 5.5e)
rc=0
check() {  # file, name fragment
  line=$(grep -F "$2" /tmp/kres_$1.txt | head -1)
  if [ -z "$line" ]; then echo "MISSING  $2"; rc=1; return; fi
  s=$(echo "$line" | sed -E 's/.*ScratchSize \[bytes\/lane\]: ([0-9]+).*/\1/')
  if [ "$s" != "0" ]; then echo "SCRATCH $s B  $2"; rc=1; else echo "ok       $2"; fi
}
for f in ncde_fast ncde_fast_fwd3 ncde_fast64 ncde_adaptive_fast; do table $C/$f.hip > /tmp/kres_$f.txt; done
check ncde_fast_fwd3 "ncde_fwd_fast_bf3ILi32ELi32ELi20ELi4ELi0ELi2ELi0ELi3ELi1ELi0E"
check ncde_fast "ncde_adj_fast3ILi3ELi20ELi0ELi2ELi0ELi0ELi2ELi0E"
check ncde_fast_fwd3 "ncde_fwd_fast_bf3ILi64ELi64ELi4ELi4ELi1ELi1ELi0ELi3ELi1ELi0E"
check ncde_fast64 "ncde_adj_h64ILi1ELi1ELi1ELi0ELi0E"
check ncde_adaptive_fast "ncde_dpf_fwdILi32ELi32ELi20E"
check ncde_adaptive_fast "ncde_dpf_tapeILi32ELi32ELi20ELi3E"
exit $rc
