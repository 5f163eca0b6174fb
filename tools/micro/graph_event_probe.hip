// Probe: do external event record / wait nodes work inside a captured HIP graph on this runtime?  (ROCm 7.2.0 image: the external
// event RECORD is captured, hipStreamWaitEvent(..., hipEventWaitExternal) on the capturing stream segfaults inside the runtime - which is why the speculative stereo match cannot be made part of the frame graph.)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/micro/graph_event_probe tools/micro/graph_event_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(int *p, int n) { int v = 0; for (int i = 0; i < n; i++) v += __builtin_amdgcn_s_memtime() & 1; if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1 + (v & 0)); }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    int *d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    hipStream_t s, other; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
    hipEvent_t mid, ext, t0, t1, t2; CK(hipEventCreateWithFlags(&mid, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ext, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreate(&t2));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, d, 2000);
    printf("capture: kernel 1 ok\n"); fflush(stdout);
    CK(hipEventRecordWithFlags(mid, s, hipEventRecordExternal));
    printf("capture: external event record ok\n"); fflush(stdout);
    CK(hipStreamWaitEvent(s, ext, hipEventWaitExternal));
    printf("capture: external event wait ok\n"); fflush(stdout);
    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, d, 2000);
    CK(hipStreamEndCapture(s, &g));
    printf("capture ended\n"); fflush(stdout);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("instantiated\n"); fflush(stdout);
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("graph nodes: %zu\n", nn);
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, other, d, rep == 4 ? 200000 : 2000);      // last repetition: the external event fires late
        CK(hipEventRecord(ext, other));
        const double a = now_us();
        CK(hipGraphLaunch(ge, s));
        int polls = 0; while (hipEventQuery(mid) == hipErrorNotReady) polls++;
        const double b = now_us();
        CK(hipStreamSynchronize(s));
        const double c = now_us();
        int h = 0; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
        printf("rep %d: mid event visible after %.1f us (%d polls), graph done after %.1f us, counter %d\n", rep, b - a, polls, c - a, h);
    }
    return 0;
}
