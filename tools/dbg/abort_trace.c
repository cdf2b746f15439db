/* debugging aid: LD_PRELOAD this (and run pytest with -p no:faulthandler) to see where a SIGABRT comes from: sender pid / code and the C backtrace of
   the thread that takes it, plus an interposed abort() for callers that go through the PLT.
   gcc -shared -fPIC -O1 -o tools/dbg/libabort_trace.so tools/dbg/abort_trace.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <fcntl.h>
static int out_fd(void) { const char *p = getenv("ABORT_TRACE_FILE"); int fd = p ? open(p, O_WRONLY | O_CREAT | O_APPEND, 0644) : -1; return fd >= 0 ? fd : 2; }   /* pytest captures fd 2 */
static void on_abrt(int sig, siginfo_t *si, void *uc)
{
    char buf[256];
    int n = snprintf(buf, sizeof buf, "\n==== SIGABRT: si_code=%d si_pid=%d si_uid=%d my_pid=%d tid=%ld ====\n", si->si_code, (int)si->si_pid, (int)si->si_uid, (int)getpid(), (long)syscall(SYS_gettid));
    const int fd = out_fd();
    (void)!write(fd, buf, n);
    void *bt[64];
    int k = backtrace(bt, 64);
    backtrace_symbols_fd(bt, k, fd);
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}
__attribute__((constructor)) static void init(void)
{
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abrt; sa.sa_flags = SA_SIGINFO;
    sigaction(SIGABRT, &sa, NULL);
}
void abort(void)
{
    static const char msg[] = "\n==== abort() called through the PLT ====\n";
    const int fd = out_fd();
    (void)!write(fd, msg, sizeof msg - 1);
    { void *bt[64]; int k = backtrace(bt, 64); backtrace_symbols_fd(bt, k, fd); }
    void (*real)(void) = (void (*)(void))dlsym(RTLD_NEXT, "abort");
    if (real) real();
    _exit(134);
}
