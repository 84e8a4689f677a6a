/* rtc_worker.c -- the compilation worker of libkabc_hip.so (capi_plugin.hip: rtc_kernel_try).
 *
 *   kabc_rtc_worker <path of libkabc_hip.so> <job file>
 *
 * Started with posix_spawn by a process that must not wait for hipRTC (the default path of a
 * model that can be specialised).  Detaches first -- fork, the first process exits at once so the
 * parent's waitpid returns immediately and nothing of the host application's child handling ever
 * sees the compilation -- then loads the library and hands it the job.  Plain C, no GPU call: the
 * compiler needs no device. */
#include <dlfcn.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <unistd.h>

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s <libkabc_hip.so> <job file>   (started by libkabc_hip.so itself)\n", argv[0]);
        return 2;
    }
    /* nothing of the parent's stays open in here (device nodes, sockets, its files): a compiler
     * that runs for seconds must not keep them alive past the parent's own close() */
    {
        long maxfd = sysconf(_SC_OPEN_MAX);
        if (maxfd < 0 || maxfd > 65536) maxfd = 65536;
        for (int fd = 3; fd < (int)maxfd; ++fd) (void)close(fd);
    }
    const pid_t p = fork();
    if (p < 0) return 3;
    if (p > 0) _exit(0);
    (void)setsid();
    signal(SIGHUP, SIG_IGN);
    if (nice(5) == -1) { /* (the sampler's host thread goes first; not being allowed to is fine) */ }
    void* dl = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!dl) return 4;
    int32_t (*run)(const char*) = (int32_t(*)(const char*))dlsym(dl, "kabc_rtc_worker_main");
    if (!run) return 5;
    /* (no dlclose, no atexit work of the compiler's statics: leave as soon as the files are written) */
    _exit(run(argv[2]));
}
