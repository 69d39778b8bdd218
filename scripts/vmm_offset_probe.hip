// Does hipMemMap accept a non-zero offset into a physical allocation handle (mapping 2-MiB slices of one handle at chosen virtual
// addresses)?  If so a buffer could be assembled from slices of two runs lying in different pieces of the device memory.
//   hipcc --offload-arch=gfx950 scripts/vmm_offset_probe.hip -o /tmp/vmm && /tmp/vmm
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(double *p, size_t n, double v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v + (double)i; }
int main()
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity %zu\n", gran);
    const size_t slice = gran < (2u << 20) ? (2u << 20) : gran, total = 8 * slice;
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, total, &prop, 0));
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    // map the slices in REVERSED order: virtual slice k <- physical slice 7 - k
    for (int k = 0; k < 8; k++) {
        hipError_t e = hipMemMap((char *)va + k * slice, slice, (size_t)(7 - k) * slice, h, 0);
        if (e != hipSuccess) { printf("hipMemMap with offset %zu -> %s\n", (size_t)(7 - k) * slice, hipGetErrorString(e)); return 2; }
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    fill<<<(unsigned)((total / 8 + 255) / 256), 256>>>((double *)va, total / 8, 1.0);
    CK(hipDeviceSynchronize());
    double x[2];
    CK(hipMemcpy(x, va, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(x + 1, (char *)va + 7 * slice, 8, hipMemcpyDeviceToHost));
    printf("OK: non-zero offsets accepted; first %g, first of slice 7 %g\n", x[0], x[1]);
    return 0;
}
