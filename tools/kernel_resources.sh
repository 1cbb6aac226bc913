#!/bin/bash
# VGPRs / scratch / LDS of every kernel in a HIP source (cross-compiled, no GPU needed). Usage: tools/kernel_resources.sh file.hip [filter]
F=$1; FILTER=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c $F -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|LDS Size" | sed 's/.*remark: [^ ]* //;s/ \[-Rpass.*//' | paste - - - - \
 | sed 's/Function Name: _ZN12_GLOBAL__N_1[0-9]*//;s/EEvN3lec9RowParamsE//;s/ScratchSize \[bytes\/lane\]/scratch/;s/LDS Size \[bytes\/block\]/lds/' | grep -E "$FILTER"
