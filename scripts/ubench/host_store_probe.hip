// Development aid: can the host store straight into device memory while a kernel that fills the chip runs?
// (rt_cancel's hipMemcpyAsync is done by a blit KERNEL for small sizes, which finds no wave slot beside the persistent
// trace kernel: profiles/r03/cancel_probe.txt.)  Tries fine-grained device memory and plain hipMalloc memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) spin(volatile unsigned int *flag, unsigned long long *out, unsigned long long limit)
{
	// every wave keeps 128 VGPRs busy so that nothing else fits beside the grid
	float acc[96];
	for (int i = 0; i < 96; i++) acc[i] = (float) (threadIdx.x + i);
	const unsigned long long t0 = wall_clock64();
	unsigned long long t = t0;
	while (true) {
		for (int r = 0; r < 64; r++)
			for (int i = 0; i < 96; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
		t = wall_clock64();
		if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
		if (t - t0 > limit) break;
	}
	float s = 0; for (int i = 0; i < 96; i++) s += acc[i];
	if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t - t0; out[1] = (unsigned long long) s; }
}
int main()
{
	int rate = 0; CHECK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));     // kHz
	for (int kind = 0; kind < 3; kind++) {
		unsigned int *flag = nullptr; unsigned long long *out = nullptr;
		hipError_t e = kind == 0 ? hipExtMallocWithFlags((void**) &flag, 4096, hipDeviceMallocFinegrained)
		             : kind == 1 ? hipMalloc((void**) &flag, 4096)
		             :             hipHostMalloc((void**) &flag, 4096, hipHostMallocMapped | hipHostMallocCoherent);
		if (e != hipSuccess) { printf("kind %d: allocation failed: %s\n", kind, hipGetErrorString(e)); continue; }
		CHECK(hipMalloc((void**) &out, 64));
		CHECK(hipMemset(flag, 0, 4096)); CHECK(hipDeviceSynchronize());
		hipPointerAttribute_t at; CHECK(hipPointerGetAttributes(&at, flag));
		printf("kind %d (%s): device ptr %p host ptr %p\n", kind, kind == 0 ? "fine-grained device" : kind == 1 ? "hipMalloc" : "pinned host, mapped", at.devicePointer, at.hostPointer);
		if (kind == 1) {          // no host pointer: try the copy routes instead, while the grid holds every wave slot
			hipStream_t s2; CHECK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, -1));
			static unsigned int one[8192]; for (auto &w : one) w = 1;
			unsigned int *pinned; CHECK(hipHostMalloc((void**) &pinned, 65536)); for (int i = 0; i < 16384; i++) pinned[i] = 1;
			unsigned int *big; CHECK(hipMalloc((void**) &big, 65536));
			for (int route = 0; route < 3; route++) {
				CHECK(hipMemset(big, 0, 65536)); CHECK(hipDeviceSynchronize());
				hipLaunchKernelGGL(spin, dim3(256 * 4), dim3(256), 0, 0, big, out, (unsigned long long) rate * 300);   // gives up after 300 ms
				std::this_thread::sleep_for(std::chrono::milliseconds(20));
				const size_t bytes = route == 0 ? 4 : (route == 1 ? 8192 : 65536);
				CHECK(hipMemcpyAsync(big, pinned, bytes, hipMemcpyHostToDevice, s2));
				CHECK(hipDeviceSynchronize());
				unsigned long long res[2]; CHECK(hipMemcpy(res, out, 16, hipMemcpyDeviceToHost));
				printf("  hipMemcpyAsync of %zu bytes from pinned memory on a high-priority stream: the grid ran %.2f ms (asked to stop after 20)\n", bytes, (double) res[0] / rate);
			}
			continue;
		}
		volatile unsigned int *h = (volatile unsigned int *) (at.hostPointer ? at.hostPointer : (void*) flag);
		hipLaunchKernelGGL(spin, dim3(256 * 4), dim3(256), 0, 0, flag, out, (unsigned long long) rate * 300);
		std::this_thread::sleep_for(std::chrono::milliseconds(20));
		*h = 1u;                   // a plain store from the CPU
		CHECK(hipDeviceSynchronize());
		unsigned long long res[2]; CHECK(hipMemcpy(res, out, 16, hipMemcpyDeviceToHost));
		printf("  CPU store: the grid ran %.2f ms (asked to stop after 20)\n", (double) res[0] / rate);
	}
	return 0;
}
