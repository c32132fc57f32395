// Poor man's sampling profiler for the host side of the library (no perf / gdb in the image):
//   gcc -O2 -fPIC -shared -o tools/prof/libsigprof.so tools/prof/sigprof.c -ldl
//   LD_PRELOAD=tools/prof/libsigprof.so GA_SIGPROF_OUT=gpurun_out/prof.txt python tools/run_configs.py 4 10 4096
// SIGPROF every 0.5 ms of process CPU time; the handler records the innermost frames inside libgraphaudio_hip.so; at exit
// the stacks are written as offsets; tools/prof/report.py turns them into per-function histograms (addr2line).
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#define MAXS 200000
#define DEPTH 24
static void* samples[MAXS][DEPTH];
static int depth[MAXS];
static volatile int nsamples = 0;

static void handler(int sig) {
  (void)sig;
  int i = __sync_fetch_and_add(&nsamples, 1);
  if (i >= MAXS) return;
  depth[i] = backtrace(samples[i], DEPTH);
}

__attribute__((constructor)) static void start(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = handler;
  sa.sa_flags = SA_RESTART;
  sigaction(SIGPROF, &sa, NULL);
  void* warm[4];
  backtrace(warm, 4);   // (loads libgcc outside the handler)
  if (getenv("GA_SIGPROF_DEFER")) return;   // the script calls sigprof_start() once the graph is built
  struct itimerval it = {{0, 500}, {0, 500}};
  setitimer(ITIMER_PROF, &it, NULL);
}
void sigprof_start(void) {
  struct itimerval it = {{0, 200}, {0, 200}};
  setitimer(ITIMER_PROF, &it, NULL);
}

// (python processes that hold a GPU context leave through _exit: call sigprof_dump() from the script -- ctypes.CDLL(None).sigprof_dump())
void sigprof_dump(void) {
  struct itimerval it = {{0, 0}, {0, 0}};
  setitimer(ITIMER_PROF, &it, NULL);
  if (nsamples == 0) return;
  const char* out = getenv("GA_SIGPROF_OUT");
  const char* lib = getenv("GA_SIGPROF_LIB") ? getenv("GA_SIGPROF_LIB") : "libgraphaudio_hip";   // (file name of the library whose frames count)
  FILE* f = fopen(out ? out : "sigprof.txt", "w");
  if (!f) return;
  int n = nsamples < MAXS ? nsamples : MAXS;
  fprintf(f, "# %d samples (0.5 ms of CPU each); per line: offsets inside libgraphaudio_hip.so, innermost first (tools/prof/report.py)\n", n);
  for (int i = 0; i < n; i++) {
    int any = 0;
    for (int d = 2; d < depth[i]; d++) {   // (0, 1: the handler and the signal trampoline)
      Dl_info di;
      if (!dladdr(samples[i][d], &di) || !di.dli_fname || !strstr(di.dli_fname, lib)) continue;
      fprintf(f, "%s%lx", any ? " " : "", (unsigned long)((char*)samples[i][d] - (char*)di.dli_fbase));
      any = 1;
    }
    fprintf(f, any ? "\n" : "-\n");
  }
  fclose(f);
  nsamples = 0;
}
__attribute__((destructor)) static void stop(void) { sigprof_dump(); }
