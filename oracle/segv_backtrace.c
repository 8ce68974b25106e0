/* LD_PRELOAD helper: print a backtrace on SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL (for crashes that happen
 * once in many runs), then the thread's name, the faulting instruction pointer and the process map, so that every
 * frame can be attributed to a library offline.  The process then ends with status 128 + signal.
 *   built by oracle/Makefile as oracle/segv_backtrace.so; tests/test_reference_programs.py and
 *   tools/crash_soak.py preload it into the reference's client programs.  TEST INFRASTRUCTURE ONLY. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>
static void put(const char * s, int len)
{
  if (write(2, s, (size_t)len) < 0) {}
}
static void handler(int sig, siginfo_t * si, void * uc)
{
  void * frames[64];
  int n = backtrace(frames, 64);
  char line[256], name[32] = "?";
  int len, fd;
  void * ip = NULL;
#if defined(__x86_64__)
  if (uc) ip = (void *)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
#endif
  prctl(PR_GET_NAME, name, 0, 0, 0);
  len = snprintf(line, sizeof(line), "\n*** signal %d (code %d) at address %p, ip %p, thread %ld '%s'\n", sig,
                 si ? si->si_code : 0, si ? si->si_addr : NULL, ip, (long)syscall(SYS_gettid), name);
  put(line, len);
  backtrace_symbols_fd(frames, n, 2);
  if (getenv("SEGV_BACKTRACE_MAPS") && (fd = open("/proc/self/maps", O_RDONLY)) >= 0)
  {
    char buf[4096];
    ssize_t got;
    put("--- maps\n", 9);
    while ((got = read(fd, buf, sizeof(buf))) > 0) put(buf, (int)got);
    close(fd);
  }
  _exit(128 + sig);
}
__attribute__((constructor)) static void install(void)
{
  static const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
  struct sigaction sa;
  unsigned int i;
  void * warm[4];
  backtrace(warm, 4); /* (loads libgcc now: the first call allocates, which a handler must not) */
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_RESETHAND | SA_ONSTACK;
  for (i = 0; i < sizeof(sigs) / sizeof(sigs[0]); ++i) sigaction(sigs[i], &sa, NULL);
}
