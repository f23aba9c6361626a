// Device memory mapped at a chosen virtual alignment (HIP virtual memory management): does the blur's time
// depend on the page-table fragments behind its arenas?  (tools/probe_arena_vmm.py)
//   hipcc --offload-arch=gfx950 -shared -fPIC -o build/probes/libvmm_alloc.so tools/probes/vmm_alloc.hip
#include <hip/hip_runtime.h>
#include <cstdio>

struct VmmBlock {
    void *va;
    size_t size;
    hipMemGenericAllocationHandle_t handle;
};

#define TRY(x)                                                                   \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "vmm_alloc: %s -> %s\n", #x, hipGetErrorString(e_)); \
            return nullptr;                                                      \
        }                                                                        \
    } while (0)

extern "C" VmmBlock *vmm_alloc(size_t bytes, size_t va_align) {
    int dev = 0;
    TRY(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = 2u << 20;
    const size_t size = (bytes + gran - 1) / gran * gran;
    VmmBlock *b = new VmmBlock{nullptr, size, {}};
    TRY(hipMemCreate(&b->handle, size, &prop, 0));
    TRY(hipMemAddressReserve(&b->va, size, va_align, nullptr, 0));
    TRY(hipMemMap(b->va, size, 0, b->handle, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    TRY(hipMemSetAccess(b->va, size, &acc, 1));
    return b;
}

extern "C" void *vmm_ptr(VmmBlock *b) { return b ? b->va : nullptr; }
extern "C" size_t vmm_size(VmmBlock *b) { return b ? b->size : 0; }

extern "C" void vmm_free(VmmBlock *b) {
    if (!b) return;
    (void)hipMemUnmap(b->va, b->size);
    (void)hipMemRelease(b->handle);
    (void)hipMemAddressFree(b->va, b->size);
    delete b;
}
