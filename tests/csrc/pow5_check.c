/* Exhaustive pin of the Fresnel shortcut (test infrastructure, CPU only).
 *
 * The reference evaluates  (float) pow(1.0 - (double) u, 5.0)  (main.c:126-129: `pow(1.0 - u, 5.0)` handed to
 * combine()'s float parameter) with u = clamp(NoV, 0, 1) (main.c:214-216).  The HIP kernels and the oracle use
 *   x = 1.0 - (double) u;  x2 = x * x;  (float) (x2 * x2 * x)
 * instead (rt_kernels.hip, oracle/rt_oracle.c).  This program compares the two on EVERY float in [0, 1]
 * (0x3f800001 values) with this machine's libm and prints the number of mismatches.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t first, last; uint64_t bad; uint32_t example; } Job;

static void *run(void *arg)
{
	Job *j = arg;
	for (uint32_t b = j->first; b != j->last; b++) {
		float u; memcpy(&u, &b, 4);
		const float want = (float) pow(1.0 - (double) u, 5.0);
		const double x = 1.0 - (double) u, x2 = x * x;
		const float got = (float) (x2 * x2 * x);
		uint32_t wb, gb; memcpy(&wb, &want, 4); memcpy(&gb, &got, 4);
		if (wb != gb) { j->bad++; j->example = b; }
	}
	return NULL;
}

int main(int argc, char **argv)
{
	int threads = argc > 1 ? atoi(argv[1]) : 8;
	if (threads < 1) threads = 1;
	if (threads > 64) threads = 64;
	const uint32_t total = 0x3f800000u + 1u;          /* +0.0f .. 1.0f inclusive */
	pthread_t th[64]; Job jobs[64];
	for (int t = 0; t < threads; t++) {
		jobs[t].first = (uint32_t) ((uint64_t) total * t / threads);
		jobs[t].last  = (uint32_t) ((uint64_t) total * (t + 1) / threads);
		jobs[t].bad = 0; jobs[t].example = 0;
		pthread_create(&th[t], NULL, run, &jobs[t]);
	}
	uint64_t bad = 0; uint32_t example = 0;
	for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); bad += jobs[t].bad; if (jobs[t].bad) example = jobs[t].example; }
	printf("%u %llu %08x\n", total, (unsigned long long) bad, example);
	return bad != 0;
}
