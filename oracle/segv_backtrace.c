/* LD_PRELOAD helper: print a backtrace on SIGSEGV / SIGBUS / SIGABRT (for crashes that happen once in many runs)
 *   built by oracle/Makefile as oracle/segv_backtrace.so; tests/test_reference_programs.py preloads it into the client programs */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig, siginfo_t * si, void * uc)
{
  void * frames[64];
  int n = backtrace(frames, 64);
  char line[128];
  int len = snprintf(line, sizeof(line), "\n*** signal %d at address %p\n", sig, si ? si->si_addr : NULL);
  (void)uc;
  if (write(2, line, (size_t)len) < 0) {}
  backtrace_symbols_fd(frames, n, 2);
  _exit(128 + sig);
}
__attribute__((constructor)) static void install(void)
{
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, NULL);
  sigaction(SIGBUS, &sa, NULL);
}
