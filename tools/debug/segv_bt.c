/* LD_PRELOAD helper: print a native backtrace on SIGSEGV / SIGABRT (debugging aid;
 * run pytest with -p no:faulthandler so that Python does not replace the handler).
 *   gcc -O1 -g -shared -fPIC -o tools/debug/segv_bt.so tools/debug/segv_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* info, void* ctx)
{
  void* frames[64];
  const int n = backtrace(frames, 64);
  char msg[128];
  const int len = snprintf(msg, sizeof msg, "\n== signal %d at address %p ==\n", sig,
                           info ? info->si_addr : 0);
  (void) !write(2, msg, len);
  backtrace_symbols_fd(frames, n, 2);
  (void) ctx;
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void)
{
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
