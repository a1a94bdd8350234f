/*
 * TEST INFRASTRUCTURE ONLY (oracle/): stopwatch around the REFERENCE's two front-half LCU functions, for bench.py's
 * cpu_baseline leg ("kind": "reference").  Linked into oracle/_ref/libsvtref.so only; the --wrap interposers of
 * ref_harness_me_dump.c / ref_harness_ois_dump.c call svt_ref_front_time_{begin,end} around the real functions.
 *
 * When SVT_REF_FRONT_TIME names a file, the per-thread CPU time (CLOCK_THREAD_CPUTIME_ID) spent inside
 * MotionEstimateLcu (Codec/EbMotionEstimation.c:3671) and OpenLoopIntraSearchLcu (:5053) of a real encoder run is summed
 * over all threads and written at exit as one JSON line: nanoseconds and calls per function.  The sum of thread time
 * divided by the LCUs per picture is the ONE-CORE cost of the front half per picture on this host, whatever the thread
 * count of the run.  Without the variable the hooks cost one relaxed load.
 *
 * Contains no reference source.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static int g_state; /* 0 unknown, 1 on, -1 off */
static const char *g_path;
static uint64_t g_ns[2], g_calls[2];
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static void report(void)
{
    FILE *f = fopen(g_path, "w");
    if (!f)
        return;
    fprintf(f, "{\"me_ns\": %llu, \"me_calls\": %llu, \"ois_ns\": %llu, \"ois_calls\": %llu}\n", (unsigned long long)g_ns[0],
            (unsigned long long)g_calls[0], (unsigned long long)g_ns[1], (unsigned long long)g_calls[1]);
    fclose(f);
}

static inline uint64_t now_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

uint64_t svt_ref_front_time_begin(void)
{
    if (__atomic_load_n(&g_state, __ATOMIC_RELAXED) == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            g_path = getenv("SVT_REF_FRONT_TIME");
            if (g_path)
                atexit(report);
            __atomic_store_n(&g_state, g_path ? 1 : -1, __ATOMIC_RELEASE);
        }
        pthread_mutex_unlock(&g_lock);
    }
    return g_state > 0 ? now_ns() : 0;
}

void svt_ref_front_time_end(int which, uint64_t t0)
{
    if (g_state <= 0)
        return;
    const uint64_t dt = now_ns() - t0;
    __atomic_add_fetch(&g_ns[which], dt, __ATOMIC_RELAXED);
    __atomic_add_fetch(&g_calls[which], 1, __ATOMIC_RELAXED);
}
