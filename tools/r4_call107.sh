# tools/micro/pingpong.hip: serial staging / matrix phases against ping-pong wave groups
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/pingpong.hip -o /tmp/pingpong 2>&1 | grep -i error
timeout 120 /tmp/pingpong > gpurun_out/r4_pingpong_full.txt 2>&1; grep real gpurun_out/r4_pingpong_full.txt
echo "== ping-pong hand-over with __syncthreads() (-DLDSBAR=0: waits for the global loads too)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DLDSBAR=0 tools/micro/pingpong.hip -o /tmp/pingpong_b 2>&1 | grep -i error
timeout 120 /tmp/pingpong_b | grep real
echo "== consecutive MFMAs on different accumulator tiles (-DINTERLEAVE=1)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DINTERLEAVE=1 tools/micro/pingpong.hip -o /tmp/pingpong_i 2>&1 | grep -i error
timeout 120 /tmp/pingpong_i | grep real
echo "== software-pipelined fragment reads in the matrix phase (-DPREFM=1)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DPREFM=1 tools/micro/pingpong.hip -o /tmp/pingpong_p 2>&1 | grep -i error
timeout 120 /tmp/pingpong_p | grep real
