#!/bin/bash
# round 6: did the rare-path additions of this round (finish_draw: the staged metric) move the 1000-dim kernel?  The kernel of the round's first commit
# (old kernels.hip against today's engine_types.h) against today's, same box, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in w1nv8_old w1nv8_new; do
    echo -n "$lib: "; NUTPIE_HIP_LIB=scratch/libs/$lib.so python scratch/ab.py "run(1000, 1024, False, E=2048, steps=20, warm=30)" 2>&1 | grep "^dim" | cut -c1-120
  done
done
