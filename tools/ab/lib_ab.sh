#!/bin/bash
# A/B of two builds of the library on one box:  tools/ab/lib_ab.sh <a.so> <b.so> -- <command...>   (run from the repo root)
a=$1; b=$2; shift 3
lib=advchain_amd/csrc/libadvchain_hip.so
cp $lib /tmp/lib_keep.so
for v in $a $b; do
  cp $v $lib
  echo "== $v"
  "$@"
done
cp /tmp/lib_keep.so $lib
