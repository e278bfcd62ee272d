/* LD_PRELOAD shim: a backtrace on SIGABRT (the HSA runtime aborts on a GPU memory fault; pytest swallows its message unless run
   with -s).  gcc -O1 -fPIC -shared tools/abrt_backtrace.c -o tools/_build/libabrt.so; LD_PRELOAD=$PWD/tools/_build/libabrt.so python -m pytest ... -s */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_abort(int sig) {
  void *frames[64];
  const char msg[] = "\n==== SIGABRT backtrace (tools/abrt_backtrace.c) ====\n";
  (void)!write(2, msg, sizeof msg - 1);
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void init(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_abort;
  sigaction(SIGABRT, &sa, 0);
}
