/* LD_PRELOAD diagnostic: print a native backtrace when the process receives SIGABRT / SIGSEGV / SIGBUS, then die the default way.
 *   gcc -shared -fPIC -O1 -o gpurun_out/abtrace.so tools/abtrace.c
 *   LD_PRELOAD=$PWD/gpurun_out/abtrace.so python -m pytest -p no:faulthandler ...
 * (lab equipment for tools/repro_abort.sh; not part of the product) */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <stdlib.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_fatal(int sig)
{
    void *bt[64];
    /* pytest captures fd 2 while a test runs: the trace goes to $ABTRACE_OUT (appended) when that is set */
    const char *path = getenv("ABTRACE_OUT");
    int fd = path ? open(path, O_WRONLY | O_CREAT | O_APPEND, 0644) : 2;
    if (fd < 0) fd = 2;
    const char *m = sig == SIGABRT ? "\n[abtrace] SIGABRT\n" : sig == SIGSEGV ? "\n[abtrace] SIGSEGV\n" : "\n[abtrace] SIGBUS\n";
    (void)!write(fd, m, strlen(m));
    int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, fd);
    /* the mappings, so that offsets in stripped libraries can be resolved afterwards */
    FILE *f = fopen("/proc/self/maps", "r");
    if (f) {
        char line[512];
        while (fgets(line, sizeof line, f))
            if (strstr(line, "r-xp") && (strstr(line, "hip") || strstr(line, "hsa") || strstr(line, "ffcnn"))) (void)!write(fd, line, strlen(line));
        fclose(f);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void abtrace_init(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_fatal;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGABRT, &sa, NULL);
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
}
