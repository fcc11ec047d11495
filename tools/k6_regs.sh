#!/bin/bash
# register / spill / scratch summary of mlp_fused2_kernel<3,1> for a set of -D flags (compile only, ~25 s)
cd /root/repo/transhuman_amd
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../include -Icsrc "$@" -c /tmp/k6/one.hip -o /tmp/k6/one.o -save-temps=obj 2>&1 | grep -E "error" | head
S=/tmp/k6/one-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|agpr_count" $S | tr '\n' ' '; echo
awk '/^_Z17mlp_fused2_kernelILi3ELi1EEv11FusedParams:/,/s_endpgm/' $S > /tmp/k6/one_k.s
echo "scratch instrs: $(grep -c scratch_ /tmp/k6/one_k.s)  accvgpr r/w/mov: $(grep -c v_accvgpr_read /tmp/k6/one_k.s)/$(grep -c v_accvgpr_write /tmp/k6/one_k.s)/$(grep -c v_accvgpr_mov /tmp/k6/one_k.s)"
