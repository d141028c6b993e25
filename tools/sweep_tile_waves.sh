#!/bin/bash
# waves per CU vs stripe-tile size (fused mode, C4, 256 spp): is a small tile better served by fewer, less contended waves?
for w in 4 6 8 10 12 16; do
  echo "== HRT_TRAVERSE_BLOCKS_PER_CU=$w"
  HRT_TRAVERSE_BLOCKS_PER_CU=$w timeout -k 10 120 python tools/tile_scaling.py 256 || exit 1
done
