#!/bin/bash
# tools/ab_libs.sh <script.py> <lib.so>...  -- runs the script once per library build (same box), in-tree build last
s=$1; shift
for l in "$@"; do echo "== $l"; MGNET_HIP_LIB=$PWD/$l python3 $s 2>&1 | tail -12; done
echo "== in-tree"; python3 $s 2>&1 | tail -12
