#!/bin/bash
# where does a kernel spill?  scratch ops between consecutive barriers:  tools/spillmap.sh <file.hip> <mangled kernel name> [extra flags]
F=$1; K=$2; shift 2
D=$(mktemp -d); cd $D
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -save-temps -Rpass-analysis=kernel-resource-usage "$@" -c $F -o x.o 2>&1 | grep -A8 "Function Name: $K" | grep -E "VGPRs:|VGPRs Spill|ScratchSize"
S=$(ls *gfx950.s); L=$(grep -n "^$K:" $S | cut -d: -f1)
awk -v s=$L 'NR>=s' $S | awk '/s_endpgm/{print; exit} {print}' > k.s
grep -n "s_barrier\|scratch_" k.s | awk -F'[:\t ]+' '{print $1, $2}' | awk '{ if ($2=="s_barrier") {print "  barrier at line", $1, "- scratch ops since previous:", n+0; n=0} else n++ } END {print "  after last barrier:", n+0}'
cp k.s /tmp/asm/last_kernel.s; cd /; rm -rf $D
