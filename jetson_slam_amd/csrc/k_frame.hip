// k_frame.hip - the host-side unpacking the reference's Frame constructor does after the front-end, moved to the device
// (SURVEY.md 8f row n4):
//   Frame.cpp:119-196  four blocking SyncedMem::to_cpu() + a host loop turning the keypoint SoA into cv::KeyPoint records
//                      -> k_unpack_keypoints: one AoS record (the memory layout of cv::KeyPoint) per keypoint, so the frame comes
//                         back with one copy for the keypoints and one for the descriptors
//   Frame.cpp:463-479, 696-706  AssignFeaturesToGrid / PosInGrid: per keypoint cell = (round((x - minX) * invW),
//                      round((y - minY) * invH)), appended to mGrid[cx][cy] in keypoint order
//                      -> k_assign_grid: CSR over cols x rows cells (cell (i, j) at i*rows + j like mGrid[i][j]), items of a cell in
//                         ascending keypoint order (the reference's push_back order)
#include "jsorb_launch.h"

namespace jsorb {

__global__ __launch_bounds__(256) void k_unpack_keypoints(const int32_t *__restrict__ soa, int n, jsorb_keypoint *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    jsorb_keypoint k;
    k.x = (float)soa[i];                                    // mvKeys[i].pt.x = kp_x[i]  (int -> float)
    k.y = (float)soa[n + i];
    k.response = (float)soa[2 * (size_t)n + i];
    k.angle = __int_as_float(soa[3 * (size_t)n + i]);       // the angle block holds float bits
    k.octave = soa[4 * (size_t)n + i];
    k.size = (float)soa[5 * (size_t)n + i];
    k.class_id = -1;                                        // cv::KeyPoint default
    out[i] = k;
}

// one workgroup: histogram -> exclusive scan -> placement -> per-cell insertion sort (cells hold a handful of keypoints)
__global__ __launch_bounds__(1024) void k_assign_grid(const int32_t *__restrict__ soa, int n, float min_x, float min_y, float inv_w, float inv_h,
                                                       int cols, int rows, int32_t *__restrict__ cell_start, int32_t *__restrict__ cell_items)
{
    extern __shared__ int s_grid[];          // [n_cells] counts -> starts, [n_cells] cursors, [1024] scan scratch
    const int n_cells = cols * rows;
    int *s_cnt = s_grid, *s_cur = s_grid + n_cells, *s_scan = s_grid + 2 * n_cells;
    const int tid = threadIdx.x;
    for (int c = tid; c < n_cells; c += 1024) s_cnt[c] = 0;
    __syncthreads();
    auto cell_of = [&](int i) -> int {
        const float x = (float)soa[i], y = (float)soa[n + i];
        const int px = (int)roundf((x - min_x) * inv_w), py = (int)roundf((y - min_y) * inv_h);      // PosInGrid
        if (px < 0 || px >= cols || py < 0 || py >= rows) return -1;
        return px * rows + py;
    };
    for (int i = tid; i < n; i += 1024) {
        const int c = cell_of(i);
        if (c >= 0) atomicAdd(&s_cnt[c], 1);
    }
    __syncthreads();
    // exclusive scan of the counts: each thread owns a contiguous chunk of cells
    const int chunk = (n_cells + 1023) / 1024;
    const int c0 = min(tid * chunk, n_cells), c1 = min(c0 + chunk, n_cells);
    int sum = 0;
    for (int c = c0; c < c1; c++) sum += s_cnt[c];
    s_scan[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int run = s_scan[tid] - sum;
    for (int c = c0; c < c1; c++) {
        const int k = s_cnt[c];
        s_cnt[c] = run;                       // start of the cell
        s_cur[c] = run;
        cell_start[c] = run;
        run += k;
    }
    if (tid == 1023) cell_start[n_cells] = s_scan[1023];
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int c = cell_of(i);
        if (c >= 0) cell_items[atomicAdd(&s_cur[c], 1)] = i;
    }
    __syncthreads();
    __threadfence_block();
    for (int c = tid; c < n_cells; c += 1024) {       // ascending keypoint order inside every cell
        const int b = s_cnt[c], e = s_cur[c];
        for (int a = b + 1; a < e; a++) {
            const int v = cell_items[a];
            int k = a - 1;
            while (k >= b && cell_items[k] > v) { cell_items[k + 1] = cell_items[k]; k--; }
            cell_items[k + 1] = v;
        }
    }
}

void launch_unpack_keypoints(const int32_t *soa, int n, jsorb_keypoint *out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unpack_keypoints, dim3((n + 255) / 256), dim3(256), 0, s, soa, n, out);
}

void launch_assign_grid(const int32_t *soa, int n, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                        int32_t *cell_start, int32_t *cell_items, hipStream_t s)
{
    const size_t lds = (size_t)(2 * cols * rows + 1024) * sizeof(int);
    hipLaunchKernelGGL(k_assign_grid, dim3(1), dim3(1024), lds, s, soa, n, min_x, min_y, inv_w, inv_h, cols, rows, cell_start, cell_items);
}

} // namespace jsorb
