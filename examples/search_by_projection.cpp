// search_by_projection.cpp - the GPU half of ORBmatcher::SearchByProjection(Frame&, const Frame&, ...) (ORBmatcher.cpp:1673-1773) and of
// Tracking::SearchLocalPoints' frustum test (Tracking.cpp:1427-1600), written against the compat shim with the reference's call pattern:
// function-static SyncedMem<float/int/unsigned char> objects, resize() twice (capacity first, then the real count), cpu_data() fill,
// to_gpu_async() + sync_stream(), the orb_cuda:: / tracking_cuda:: entry points on gpu_data(), to_cpu_async() + sync_stream().
// Usage: search_by_projection in.bin out.bin
//   in.bin : int32 n, n_pairs, n_desc; float P[3][n], Pn[3][n], dist[3][n], R[9], t[3], Ow[3], cam[8] (fx fy cx cy minX maxX minY maxY), logsf;
//            int32 il[n_pairs], ir[n_pairs]; uint8 dl[n_desc][32], dr[n_desc][32]
//   out.bin: float u[n], v[n], invz[n]; uint8 valid[n]; int32 dist[n_pairs]; float fz[n], fu[n], fv[n], fvc[n]; int32 level[n]; uint8 in[n]
// Build: g++ -std=c++17 -I include examples/search_by_projection.cpp -L jetson_slam_amd -ljsorb
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "jsorb_compat.hpp"

using orb_cuda::SyncedMem;

static void rd(FILE *f, void *p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

static void project(int n_keypoints, int n_count, const float *P, const float *R, const float *t, float *cam, FILE *out)
{
    static SyncedMem<float> u, v, invz;
    static SyncedMem<unsigned char> is_valid;
    static SyncedMem<float> Px, Py, Pz, Rcw_smem, tcw_smem;
    // ORBmatcher.cpp:1690-1700: sized for all keypoints of the last frame first ...
    Px.resize(n_keypoints); Py.resize(n_keypoints); Pz.resize(n_keypoints);
    u.resize(n_keypoints); v.resize(n_keypoints); invz.resize(n_keypoints); is_valid.resize(n_keypoints);
    Rcw_smem.resize(9); tcw_smem.resize(3);
    float *Px_cpu = Px.cpu_data(), *Py_cpu = Py.cpu_data(), *Pz_cpu = Pz.cpu_data();
    for (int i = 0; i < 9; i++) Rcw_smem.cpu_data()[i] = R[i];
    for (int i = 0; i < 3; i++) tcw_smem.cpu_data()[i] = t[i];
    for (int i = 0; i < n_count; i++) { Px_cpu[i] = P[i]; Py_cpu[i] = P[n_count + i]; Pz_cpu[i] = P[2 * n_count + i]; }
    // ... :1733-1740 then shrunk to the valid count (no reallocation: capacity_ stays)
    Px.resize(n_count); Py.resize(n_count); Pz.resize(n_count); u.resize(n_count); v.resize(n_count); invz.resize(n_count); is_valid.resize(n_count);
    if (Px.capacity_ != n_keypoints || Px.count_ != n_count) { fprintf(stderr, "SyncedMem::resize must grow only\n"); exit(3); }
    Px.to_gpu_async(); Py.to_gpu_async(); Pz.to_gpu_async(); Rcw_smem.to_gpu_async(); tcw_smem.to_gpu_async();          // :1742-1746
    Px.sync_stream(); Py.sync_stream(); Pz.sync_stream(); Rcw_smem.sync_stream(); tcw_smem.sync_stream();               // :1748-1752
    orb_cuda::ORB_Search_by_projection_project_on_frame(n_count, Px.gpu_data(), Py.gpu_data(), Pz.gpu_data(), Rcw_smem.gpu_data(), tcw_smem.gpu_data(),
                                                        cam[0], cam[1], cam[2], cam[3], cam[4], cam[5], cam[6], cam[7],
                                                        u.gpu_data(), v.gpu_data(), invz.gpu_data(), is_valid.gpu_data());      // :1755-1763
    u.to_cpu_async(); v.to_cpu_async(); invz.to_cpu_async(); is_valid.to_cpu_async();                                    // :1767-1770
    u.sync_stream(); v.sync_stream(); invz.sync_stream(); is_valid.sync_stream();                                        // :1772-1775
    fwrite(u.cpu_data(), 4, n_count, out); fwrite(v.cpu_data(), 4, n_count, out); fwrite(invz.cpu_data(), 4, n_count, out);
    fwrite(is_valid.cpu_data(), 1, n_count, out);
}

static void distances(int n_pairs, int n_desc, const int *il, const int *ir, const unsigned char *dl, const unsigned char *dr, FILE *out)
{
    // ORBmatcher.cpp:1864-1890: index lists through SyncedMem<int>, descriptors resident on the device
    static SyncedMem<int> idx_last, idx_curr, distance;
    static SyncedMem<unsigned char> desc_l, desc_r;
    idx_last.resize(n_pairs); idx_curr.resize(n_pairs); distance.resize(n_pairs);
    desc_l.resize(32 * n_desc); desc_r.resize(32 * n_desc);
    for (int i = 0; i < n_pairs; i++) { idx_last.cpu_data()[i] = il[i]; idx_curr.cpu_data()[i] = ir[i]; }
    for (int i = 0; i < 32 * n_desc; i++) { desc_l.cpu_data()[i] = dl[i]; desc_r.cpu_data()[i] = dr[i]; }
    idx_last.to_gpu_async(); idx_curr.to_gpu_async(); desc_l.to_gpu(); desc_r.to_gpu();
    idx_last.sync_stream(); idx_curr.sync_stream();
    distance.set_zero_gpu();
    orb_cuda::ORB_compute_distances(n_pairs, idx_last.gpu_data(), idx_curr.gpu_data(), desc_l.gpu_data(), desc_r.gpu_data(), distance.gpu_data());
    distance.to_cpu();
    fwrite(distance.cpu_data(), 4, n_pairs, out);
}

static void frustum(int n_points, const float *P, const float *Pn, const float *D, const float *R, const float *t, const float *Ow_, float *cam, float logsf,
                    FILE *out)
{
    // Tracking.cpp:1427-1449, 1566-1600
    static SyncedMem<float> Px, Py, Pz, Pnx, Pny, Pnz, invz, u, v, viewCos, invariance_maxDistance, invariance_minDistance, MaxDistance, Rcw, tcw, Ow;
    static SyncedMem<int> predictedlevel;
    static SyncedMem<unsigned char> isinfrustum;
    SyncedMem<float> *in[] = {&Px, &Py, &Pz, &Pnx, &Pny, &Pnz, &MaxDistance, &invariance_maxDistance, &invariance_minDistance};
    const float *src[] = {P, P + n_points, P + 2 * n_points, Pn, Pn + n_points, Pn + 2 * n_points, D, D + n_points, D + 2 * n_points};
    for (int k = 0; k < 9; k++) {
        in[k]->resize(n_points);
        for (int i = 0; i < n_points; i++) in[k]->cpu_data()[i] = src[k][i];
        in[k]->to_gpu_async();
    }
    Rcw.resize(9); tcw.resize(3); Ow.resize(3);
    for (int i = 0; i < 9; i++) Rcw.cpu_data()[i] = R[i];
    for (int i = 0; i < 3; i++) { tcw.cpu_data()[i] = t[i]; Ow.cpu_data()[i] = Ow_[i]; }
    Rcw.to_gpu(); tcw.to_gpu(); Ow.to_gpu();
    SyncedMem<float> *outf[] = {&invz, &u, &v, &viewCos};
    for (auto *o : outf) {             // outputs carry a sentinel: the kernel writes them only where the point is in the frustum
        o->resize(n_points);
        for (int i = 0; i < n_points; i++) o->cpu_data()[i] = -7.0f;
        o->to_gpu();
    }
    predictedlevel.resize(n_points);
    for (int i = 0; i < n_points; i++) predictedlevel.cpu_data()[i] = -7;
    predictedlevel.to_gpu();
    isinfrustum.resize(n_points);
    isinfrustum.set_zero_gpu();
    for (int k = 0; k < 9; k++) in[k]->sync_stream();
    int minX = (int)cam[4], maxX = (int)cam[5], minY = (int)cam[6], maxY = (int)cam[7], nScaleLevels = 8;
    float viewCosAngle = 0.5f;
    tracking_cuda::compute_isInFrustum_GPU(n_points, Px.gpu_data(), Py.gpu_data(), Pz.gpu_data(), Pnx.gpu_data(), Pny.gpu_data(), Pnz.gpu_data(),
                                           MaxDistance.gpu_data(), invariance_maxDistance.gpu_data(), invariance_minDistance.gpu_data(), Rcw.gpu_data(),
                                           tcw.gpu_data(), Ow.gpu_data(), cam[0], cam[1], cam[2], cam[3], minX, maxX, minY, maxY, nScaleLevels, logsf,
                                           viewCosAngle, invz.gpu_data(), u.gpu_data(), v.gpu_data(), predictedlevel.gpu_data(), viewCos.gpu_data(),
                                           isinfrustum.gpu_data());
    for (auto *o : outf) o->to_cpu_async();
    predictedlevel.to_cpu_async(); isinfrustum.to_cpu_async();
    for (auto *o : outf) { o->sync_stream(); fwrite(o->cpu_data(), 4, n_points, out); }
    predictedlevel.sync_stream(); isinfrustum.sync_stream();
    fwrite(predictedlevel.cpu_data(), 4, n_points, out);
    fwrite(isinfrustum.cpu_data(), 1, n_points, out);
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    int hdr[3];
    rd(f, hdr, sizeof hdr);
    const int n = hdr[0], n_pairs = hdr[1], n_desc = hdr[2];
    std::vector<float> P(3 * n), Pn(3 * n), D(3 * n);
    float R[9], t[3], Ow[3], cam[8], logsf;
    rd(f, P.data(), 12 * n); rd(f, Pn.data(), 12 * n); rd(f, D.data(), 12 * n);
    rd(f, R, sizeof R); rd(f, t, sizeof t); rd(f, Ow, sizeof Ow); rd(f, cam, sizeof cam); rd(f, &logsf, 4);
    std::vector<int> il(n_pairs), ir(n_pairs);
    rd(f, il.data(), 4 * n_pairs); rd(f, ir.data(), 4 * n_pairs);
    std::vector<unsigned char> dl(32 * n_desc), dr(32 * n_desc);
    rd(f, dl.data(), dl.size()); rd(f, dr.data(), dr.size());
    fclose(f);
    FILE *out = fopen(argv[2], "wb");
    try {
        for (int round = 0; round < 2; round++) {      // second round: the function statics are reused (no reallocation), results must not change
            if (round == 1) { fclose(out); out = fopen(argv[2], "wb"); }
            project(n + 37, n, P.data(), R, t, cam, out);
            distances(n_pairs, n_desc, il.data(), ir.data(), dl.data(), dr.data(), out);
            frustum(n, P.data(), Pn.data(), D.data(), R, t, Ow, cam, logsf, out);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    fclose(out);
    printf("ok n=%d pairs=%d\n", n, n_pairs);
    return 0;
}
