#!/bin/bash
# usage (on the GPU box): bash tools/lib_ab.sh tools/_probe/lib_A.so tools/_probe/lib_B.so [rounds]
# (STEP_ARGS="--dtype bf16x3 --steps 40" for another path)
# alternates tools/step_ms.py between two builds of the library on ONE box (processes alternate: A B A B ...)
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  echo -n "A $(basename $A): "; SL_LIB_PATH=$A python tools/step_ms.py $STEP_ARGS 2>&1 | grep median
  echo -n "B $(basename $B): "; SL_LIB_PATH=$B python tools/step_ms.py $STEP_ARGS 2>&1 | grep median
done
