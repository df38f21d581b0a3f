// TEST INFRASTRUCTURE ONLY — see fakehip/hip/hip_runtime.h
#include <hip/hip_runtime.h>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
namespace iopx { alignas(16) uint64_t iopx_smem[160 * 1024 / 8]; }

thread_local int iopx_emu_lane = 0;

void emu_launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body)
{
    (void)block;
    if (lds_bytes > 160 * 1024) { fprintf(stderr, "emu: LDS request %zu exceeds 160 KiB\n", lds_bytes); abort(); }
    gridDim = grid;
    blockDim = dim3(1, 1, 1);
    threadIdx = dim3(0, 0, 0);
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                blockIdx = dim3(x, y, z);
                // poison LDS so that reads of unwritten slots are visible as garbage, not stale data
                memset(iopx::iopx_smem, 0xA5, lds_bytes);
                body();
            }
}
