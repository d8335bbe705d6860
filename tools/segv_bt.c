// debugging aid (no gdb on the GPU box): prints a native backtrace on SIGSEGV/SIGBUS/SIGABRT.
// LD_PRELOAD it, or dlopen it and call segv_bt_install() after the HIP runtime has installed its own handlers.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <unistd.h>
static void h(int sig) { void *b[64]; int n = backtrace(b, 64); backtrace_symbols_fd(b, n, 2); _exit(128 + sig); }
void segv_bt_install(void) { signal(SIGSEGV, h); signal(SIGBUS, h); signal(SIGABRT, h); }
__attribute__((constructor)) static void init(void) { segv_bt_install(); }
