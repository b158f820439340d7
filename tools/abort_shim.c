/* Debug aid: LD_PRELOAD this to get a C backtrace of whichever thread raises SIGABRT / SIGSEGV / SIGBUS (pytest's fd
 * capture swallows the runtime's own message).  gcc -shared -fPIC -o build/abort_shim.so tools/abort_shim.c -ldl */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int saved_err = -1;
static void handler(int sig)
{
    const char *path = getenv("ABORT_SHIM_OUT");
    int fd = open(path ? path : "/tmp/abort_bt.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
    void *bt[96];
    int n = backtrace(bt, 96);
    const char *hdr = sig == SIGABRT ? "== SIGABRT\n" : sig == SIGSEGV ? "== SIGSEGV\n" : "== SIGBUS\n";
    if (fd >= 0) {
        (void)!write(fd, hdr, strlen(hdr));
        backtrace_symbols_fd(bt, n, fd);
        /* what the process wrote to its (captured) stderr so far: the runtime's own message is in there */
        int f2 = open("/proc/self/fd/2", O_RDONLY);
        if (f2 >= 0) {
            char buf[4096];
            ssize_t k;
            (void)!write(fd, "-- captured stderr --\n", 22);
            while ((k = read(f2, buf, sizeof buf)) > 0) (void)!write(fd, buf, (size_t)k);
            close(f2);
        }
        close(fd);
    }
    if (saved_err >= 0) {
        (void)!write(saved_err, hdr, strlen(hdr));
        backtrace_symbols_fd(bt, n, saved_err);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void)
{
    saved_err = dup(2); /* the real stderr, before any capture */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sigaction(SIGABRT, &sa, NULL);
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
}
