/* debug aid: LD_PRELOAD this to get the C call stack of an abort() (glibc heap checks, assert) on boxes without gdb.
 * build: gcc -shared -fPIC -o tools/libabort_bt.so tools/abort_bt.c */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_abort(int sig) {
    void* bt[64];
    int n = backtrace(bt, 64);
    const char msg[] = "\n==== abort_bt: C stack at the fatal signal ====\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) { signal(SIGABRT, on_abort); signal(SIGSEGV, on_abort); signal(SIGBUS, on_abort); }
