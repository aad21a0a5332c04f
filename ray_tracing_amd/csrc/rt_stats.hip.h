/*
 * rt_stats.hip.h -- DEVELOPMENT instrumentation of the trace kernels; never part of the product build.
 *
 * Included by rt_kernels.hip only when RT_STATS is defined (`make stats` -> librt_hip_stats.so; a compiled scene's kernel with
 * rt_tuning.jit_flags = "-DRT_STATS ...": scripts/stats_c1.py, scripts/stats_large.py, scripts/probes/tail_probe.py).  The
 * kernels carry three kinds of hooks, all of which expand to nothing without it:
 *   STAT(site)        executions and active lanes at a source site, accumulated in the device array rt_stats
 *                     (sites 0..31: words 0..63 -- 50..57 are the section stamps' words --; 32..63: the culled trace)
 *   STAMP(k)          a wave's clock since the previous stamp is booked to section k (rt_stats[50 + k])
 *   STAMP_DRY/ROUND   wave life times (below)
 * Modes:  -DRT_STATS                            sites and stamps
 *         -DRT_STATS -DRT_STATS_STAMPS_ONLY     the per-site atomics distort the section times: this build keeps the stamps only
 *         -DRT_STATS -DRT_STATS_LIFETIMES_ONLY  the per-site atomics slow a launch down a hundredfold and the section stamps by a
 *                                               third; what a look at the end of a launch needs is the real pace: every wave writes
 *                                               four words of its own -- start, the time it found the pixel lists empty, the rounds
 *                                               it ran after that, its end (s_memrealtime, 10 ns) -- and nothing else
 */
#ifndef RT_STATS_HIP_H
#define RT_STATS_HIP_H

extern "C" { __device__ unsigned long long rt_stats[128]; }

#ifdef RT_STATS_LIFETIMES_ONLY
extern "C" { __device__ unsigned long long rt_wave_log[4 * 8192]; }
struct RtStamps {
	unsigned long long t0, tdry = 0, dry_rounds = 0;
	__device__ __forceinline__ RtStamps() { t0 = __builtin_amdgcn_s_memrealtime(); }
	__device__ __forceinline__ void mark(int) {}
	__device__ __forceinline__ void dry() { if (!tdry) tdry = __builtin_amdgcn_s_memrealtime(); }
	__device__ __forceinline__ void round() { if (tdry) dry_rounds++; }
	__device__ __forceinline__ void flush(unsigned int waves_per_block)
	{
		const unsigned int w_ = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
		if ((threadIdx.x & 63) == 0 && w_ < 8192u) {
			rt_wave_log[4 * w_] = t0; rt_wave_log[4 * w_ + 1] = tdry; rt_wave_log[4 * w_ + 2] = dry_rounds; rt_wave_log[4 * w_ + 3] = __builtin_amdgcn_s_memrealtime();
		}
	}
};
#define STAT(site) do {} while (0)
#else
struct RtStamps {
	unsigned long long sec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last;
	__device__ __forceinline__ RtStamps() { last = __builtin_amdgcn_s_memtime(); }
	__device__ __forceinline__ void mark(int k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sec[k] += now_ - last; last = now_; }
	__device__ __forceinline__ void dry() {}
	__device__ __forceinline__ void round() {}
	__device__ __forceinline__ void flush(unsigned int) { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 8; k_++) atomicAdd(&rt_stats[50 + k_], sec[k_]); }
};
#ifdef RT_STATS_STAMPS_ONLY
#define STAT(site) do {} while (0)
#else
#define STAT(site) do { const unsigned long long m_ = __ballot(true); \
	if (__builtin_amdgcn_mbcnt_hi((unsigned int) (m_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) m_, 0u)) == 0) { \
		atomicAdd(&rt_stats[2 * (site)], 1ull); atomicAdd(&rt_stats[2 * (site) + 1], (unsigned long long) __popcll(m_)); } } while (0)
#endif
#endif

#define STAMP_LOCAL  RtStamps rt_stamps
#define STAMP(k)     rt_stamps.mark(k)
#define STAMP_DRY    rt_stamps.dry()
#define STAMP_ROUND  rt_stamps.round()
#define STAMP_FLUSH(waves_per_block) rt_stamps.flush(waves_per_block)

#ifndef __HIPCC_RTC__       /* the library's own copy of the counters (librt_hip_stats.so: scripts/stats_large.py) */
extern "C" __attribute__((visibility("default"))) int rt_stats_read(unsigned long long out[128], int reset)
{
	if (hipDeviceSynchronize() != hipSuccess) return -2;
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rt_stats), 128 * sizeof(unsigned long long)) != hipSuccess) return -2;
	if (reset) {
		unsigned long long zero[128] = {0};
		if (hipMemcpyToSymbol(HIP_SYMBOL(rt_stats), zero, sizeof(zero)) != hipSuccess) return -2;
	}
	return 0;
}
#endif

#endif
