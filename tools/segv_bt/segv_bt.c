// LD_PRELOAD helper (diagnostics only, tools/segv_hunt.sh): native backtrace of the faulting thread on SIGSEGV / SIGBUS /
// SIGABRT, then the default action.  glibc backtrace() + backtrace_symbols_fd(): async-signal-unsafe in theory, good
// enough for a process that is dying anyway.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <ucontext.h>

static void handler(int sig, siginfo_t* si, void* uc_) {
  char buf[160];
  ucontext_t* uc = (ucontext_t*)uc_;
  int n = snprintf(buf, sizeof buf, "\n=== segv_bt: signal %d, fault address %p, rip %p ===\n", sig, si->si_addr,
                   (void*)uc->uc_mcontext.gregs[REG_RIP]);
  write(2, buf, n);
  void* frames[96];
  int k = backtrace(frames, 96);
  backtrace_symbols_fd(frames, k, 2);
  // /proc/self/maps lines of the libraries the frames fall into would be long; the symbol names are what is needed
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
  static char stack[1 << 16];
  stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
  sigaltstack(&ss, 0);
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
