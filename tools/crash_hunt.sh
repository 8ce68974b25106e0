#!/bin/bash
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# a crash of a reference test program seen once in a full test run: repeat it with a backtrace handler preloaded
export PLLHIP_AA_EXACT=1
n=0
for i in $(seq 1 250); do
  for t in derivatives-oddstates derivatives; do
    LD_PRELOAD=$PWD/oracle/segv_backtrace.so oracle/_ref/reftest_$t > /tmp/out 2> /tmp/err; rc=$?
    if [ $rc != 0 ]; then n=$((n+1)); echo "== $t run $i rc=$rc"; tail -40 /tmp/err; fi
  done
done
echo "crashes: $n of 500"
