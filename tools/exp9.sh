#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for L in d0 d2 d0 d2 d0 d2; do echo "### $L"; T360_LIB=$R/tools/ab/libT360_$L.so tools/sweep.sh "X=1"; done
