/* tools/segv_trace.c -- a SIGSEGV handler that prints the native backtrace (glibc backtrace_symbols_fd: module + offset per
 * frame) before the default action takes the process down.  Loaded by bench.py with ctypes when FMCMC_SEGV_TRACE=1, so that a
 * profiled run (`rocprofv3 ... -- python3 bench.py ...`, the program directly after `--`) that dies in an exit handler says
 * WHOSE frame it died in.  Diagnostic only; build: gcc -shared -fPIC -O1 tools/segv_trace.c -o tools/exp_bin/libsegv_trace.so */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static char alt_stack[1 << 16];
static int out_fd = 2;

static void on_segv(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  static const char head[] = "\n==== fmcmc segv_trace: SIGSEGV, native frames (innermost first) ====\n";
  void* frames[64];
  if (write(out_fd, head, sizeof head - 1) < 0) {}
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, out_fd);
  static const char tail[] = "==== end of frames ====\n";
  if (write(out_fd, tail, sizeof tail - 1) < 0) {}
  (void)si;
  signal(sig, SIG_DFL);
  raise(sig);
}

void fmcmc_segv_trace_install(const char* path) {
  if (path && *path) {
    int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) out_fd = fd;
  }
  void* warm[4];
  backtrace(warm, 4);                 /* loads libgcc's unwinder now: not from inside the handler */
  stack_t ss;
  ss.ss_sp = alt_stack; ss.ss_size = sizeof alt_stack; ss.ss_flags = 0;
  sigaltstack(&ss, 0);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGSEGV, &sa, 0);
}
