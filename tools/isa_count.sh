#!/bin/bash
# usage: tools/isa_count.sh <file.hip> <mangled-symbol-prefix>   -- static instruction mix of one kernel
set -e
SRC=$1; SYM=$2
TMP=/tmp/isa_$$.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only -o $TMP $SRC 2>/dev/null
A=$(grep -n "^$SYM" $TMP | head -1 | cut -d: -f1)
awk -v a=$A 'NR>=a' $TMP | awk '/s_endpgm/{exit} {print}' > /tmp/isa_body_$$.s
echo "VALU $(grep -cE '^\s+v_' /tmp/isa_body_$$.s)  SALU $(grep -cE '^\s+s_' /tmp/isa_body_$$.s)  LDS $(grep -cE '^\s+ds_' /tmp/isa_body_$$.s)  VMEM $(grep -cE '^\s+global_' /tmp/isa_body_$$.s)"
awk -v a=$A 'NR>=a' $TMP | grep -m3 "NumVgprs\|NumSgprs\|Occupancy"
rm -f $TMP /tmp/isa_body_$$.s
