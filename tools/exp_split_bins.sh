#!/bin/bash
# tools/exp_split_bins.sh OUT -- the device split builder with other bin counts (rebuilds build_split.o on the box), C4 rate for each
out=$1; mkdir -p "$(dirname "$out")"; : > "$out.log"
variant() {
  touch nvidia-optix-ray-tracer_amd/csrc/build_split.hip
  make lib EXTRA_HIPFLAGS="$1" >> "$out.log" 2>&1 || { echo "build failed: $1" >> "$out"; return; }
  shift
  tools/sweep_device_split.sh "$out.part" "$@" > /dev/null
  cat "$out.part" >> "$out"
}
echo "## obj 32" >> "$out"; variant "-DHRT_SPLIT_OBJ_BINS=32" HRT_SBVH_CELL_REFS=16
echo "## sp 64" >> "$out"; variant "-DHRT_SPLIT_SP_BINS=64" HRT_SBVH_CELL_REFS=16
echo "## default bins" >> "$out"; variant "" HRT_SBVH_CELL_REFS=4 HRT_SBVH_CELL_REFS=2
cat "$out"
