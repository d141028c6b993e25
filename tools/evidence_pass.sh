#!/bin/bash
# The round-end evidence that tools/profile_final.sh and tools/profile_two_level.sh do not take: canonical against executed counts of the two-level trees,
# the Time-mode loop, the tile projection, the refit bench, the other configurations, the bench line.  Writes under gpurun_out/final2/.
set -u
O=gpurun_out/final2; mkdir -p $O
B=$(cat nvidia-optix-ray-tracer_amd/lib/BUILD_ID)
HRT_LIB=$PWD/nvidia-optix-ray-tracer_amd/lib/libhrt_stats.so timeout -k 10 300 python3 tools/two_level_counts.py 2000 > $O/counts.txt 2>&1 || exit 1
HRT_LIB=$PWD/nvidia-optix-ray-tracer_amd/lib/libhrt_stats.so timeout -k 10 400 python3 tools/two_level_counts.py 100000 >> $O/counts.txt 2>&1 || exit 1
echo counts done
tools/profile_time_mode.sh r04 > $O/time_mode.log 2>&1 || exit 1
cp gpurun_out/final/r04_time_mode_driver.txt gpurun_out/final/r04_time_mode_kernel_stats.csv $O/
echo time mode done
{ echo "# tools/tile_scaling.py 256 on build $B: the rank-0 tile of the N-GPU split rendered alone on one MI355X (C4, 256 spp, default device build); a PROJECTION, not a multi-GPU run"; timeout -k 10 400 python3 tools/tile_scaling.py 256 2>&1 | grep -v amdgpu.ids; } > $O/r04_tile_scaling_projection.txt || exit 1
echo tiles done
{ echo "# tools/refit_bench.py on build $B (round 4; round 3: profiles/r03_device_split_build.txt)"; timeout -k 10 400 python3 tools/refit_bench.py 2>&1 | grep -v amdgpu.ids; } > $O/r04_refit_bench.txt || exit 1
echo refit done
timeout -k 10 600 tools/bench_other_configs.sh > $O/r04_bench_other_configs.txt 2>&1 || exit 1
echo configs done
timeout -k 10 400 python3 bench.py > $O/r04_bench_c4_256spp.json 2> $O/bench.err || exit 1
echo bench done
