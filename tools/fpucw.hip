// diagnostic: does the HIP runtime change the x87 control word of the calling thread?
#include <hip/hip_runtime.h>
#include <stdio.h>
static unsigned short cw() { unsigned short v; __asm__ __volatile__("fnstcw %0" : "=m"(v)); return v; }
__global__ void k(int *p) { p[threadIdx.x] = threadIdx.x; }
int main() {
    printf("start        cw=%04x\n", cw());
    int n = 0; hipGetDeviceCount(&n); printf("after count  cw=%04x n=%d\n", cw(), n);
    int *d; hipMalloc(&d, 1024); printf("after malloc cw=%04x\n", cw());
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize(); printf("after launch cw=%04x\n", cw());
    return 0;
}
