#!/bin/bash
# A/B of the random-uniform merge routes and of the rays-per-wave shapes of mvip_sample_pdf_merge (both are read once per
# process from the environment): counting merge on / off x rays per wave 1 / 2 / 4, at the frame size (190,512 rays,
# cache-resident) and at 8 frames (past the Infinity Cache).  Writes gpurun_out/r6_sample_merge_ab.jsonl.
out=gpurun_out/r6_sample_merge_ab.jsonl
: > $out
for c in 1 0; do for r in 1 2 4; do
  MVIP_SAMPLE_COUNTING=$c MVIP_SAMPLE_RPW=$r python tools/micro_bench.py 2>/dev/null | grep sample_pdf_merge | sed "s/^{/{\"counting\": $c, \"rays_per_wave\": $r, /" >> $out
done; done
cat $out
