#!/bin/bash
# round-3 session 10: streaming conv, fragment read-ahead 4 / 8 / 12 (full kernel and without any DMA)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp10; mkdir -p $O
cd $R
for L in p8; do
  SRGANFD_LIB=$R/build_exp/libsrganfd_$L.so timeout -k 10 200 python tools/kbench.py --dtype f16 --modes 8 --igv 0 --dbg 0,17 --rounds 3 --reps 10 --only "cout" > $O/kb_$L.txt 2>&1 || { tail -5 $O/kb_$L.txt; exit 1; }
  SRGANFD_LIB=$R/build_exp/libsrganfd_$L.so timeout -k 10 200 python tools/kbench.py --dtype f16 --modes 8 --igv 0 --dbg 0,17 --rounds 3 --reps 10 --set gan --only "256->256" >> $O/kb_$L.txt 2>&1 || { tail -5 $O/kb_$L.txt; exit 1; }
  echo "== $L"; grep -v amdgpu $O/kb_$L.txt | cut -c1-125
done
