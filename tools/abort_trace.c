/* LD_PRELOAD helper for the GPU box: prints the native call stack of the thread that raises SIGABRT (an abort() inside the HIP runtime or a std::terminate leaves no
   message otherwise), then lets the default action run.   built ON DEMAND, never shipped:  gcc -shared -fPIC -O1 -o /tmp/_abort_trace.so tools/abort_trace.c
   LD_PRELOAD=/tmp/_abort_trace.so python -m pytest -p no:faulthandler ...                                                                                        */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
  void* frames[96];
  const char msg[] = "\n=== abort_trace: native stack of the aborting thread ===\n";
  if (write(2, msg, sizeof(msg) - 1) < 0) {}
  int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_abort;
  sigaction(SIGABRT, &sa, 0);
}
