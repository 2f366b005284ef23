/* Diagnostic only (tests/conftest.py loads it when JEN1_ABORT_TRACE=1): on SIGABRT / SIGSEGV / SIGBUS print the NATIVE call stack of the
 * faulting thread to stderr and to $JEN1_ABORT_TRACE_FILE, then hand over to the handler that was installed before (Python's faulthandler).
 * Build: gcc -O1 -g -shared -fPIC -o tools/native/libabort_trace.so tools/native/abort_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static struct sigaction prev_abrt, prev_segv, prev_bus;
static int trace_fd = -1;

static void say(int fd, const char *s) { if (fd >= 0) { ssize_t r = write(fd, s, strlen(s)); (void)r; } }

static void on_signal(int sig, siginfo_t *info, void *ctx) {
  void *frames[96];
  int n = backtrace(frames, 96);
  const char *name = sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : "SIGBUS";
  int fds[2] = {2, trace_fd};
  for (int i = 0; i < 2; i++) {
    if (fds[i] < 0) continue;
    say(fds[i], "\n[jen1 abort_trace] native stack at ");
    say(fds[i], name);
    say(fds[i], ":\n");
    backtrace_symbols_fd(frames, n, fds[i]);
  }
  struct sigaction *prev = sig == SIGABRT ? &prev_abrt : sig == SIGSEGV ? &prev_segv : &prev_bus;
  if ((prev->sa_flags & SA_SIGINFO) && prev->sa_sigaction) { prev->sa_sigaction(sig, info, ctx); return; }
  if (!(prev->sa_flags & SA_SIGINFO) && prev->sa_handler != SIG_DFL && prev->sa_handler != SIG_IGN) { prev->sa_handler(sig); return; }
  signal(sig, SIG_DFL);
  raise(sig);
}

int jen1_abort_trace_install(const char *path) {
  void *warm[4];
  backtrace(warm, 4);                                   /* loads libgcc now, not inside the handler */
  if (path && *path) trace_fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_signal;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGABRT, &sa, &prev_abrt);
  sigaction(SIGSEGV, &sa, &prev_segv);
  sigaction(SIGBUS, &sa, &prev_bus);
  return 0;
}
