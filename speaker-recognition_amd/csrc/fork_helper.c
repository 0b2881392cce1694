/* fork_helper.c -- lib/sr_fork_helper: the process that computes for a child forked after its parent had initialised
 * the GPU runtime (fork_proxy.cpp; the reference's fit-then-Pool drivers, src/test/test-nperson.py:126-139).
 * Started by posix_spawn with one connected socket (its number is argv[1]); loads lib/pygmm.so from its own
 * directory -- a fresh address space, so a fresh HIP runtime -- and serves requests until the socket closes. */
#define _GNU_SOURCE
#include <dirent.h>
#include <dlfcn.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int main(int argc, char **argv) {
    if (argc != 2) {
        fprintf(stderr, "sr_fork_helper: started by lib/pygmm.so, not by hand (usage: sr_fork_helper <socket fd>)\n");
        return 2;
    }
    const int fd = atoi(argv[1]);
    /* the forked child's other descriptors (a multiprocessing pool's pipes, say) are none of this process's business:
     * holding their write ends open would hide the child's exit from whoever waits on them */
    DIR *d = opendir("/proc/self/fd");
    if (d) {
        int to_close[4096], n = 0;
        struct dirent *e;
        while ((e = readdir(d)) != NULL) {
            const int k = atoi(e->d_name);
            if (e->d_name[0] >= '0' && e->d_name[0] <= '9' && k > 2 && k != fd && k != dirfd(d) && n < 4096) to_close[n++] = k;
        }
        closedir(d);
        for (int i = 0; i < n; i++) close(to_close[i]);
    }
    char exe[PATH_MAX];
    const ssize_t len = readlink("/proc/self/exe", exe, sizeof exe - 1);
    if (len <= 0) {
        fprintf(stderr, "sr_fork_helper: cannot resolve /proc/self/exe\n");
        return 2;
    }
    exe[len] = 0;
    char *slash = strrchr(exe, '/');
    if (!slash || (size_t)(slash - exe) + sizeof "/pygmm.so" > sizeof exe) return 2;
    strcpy(slash, "/pygmm.so");
    void *lib = dlopen(exe, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        fprintf(stderr, "sr_fork_helper: %s\n", dlerror());
        return 2;
    }
    int (*serve)(int) = (int (*)(int))dlsym(lib, "sr_fork_helper_main");
    if (!serve) {
        fprintf(stderr, "sr_fork_helper: %s\n", dlerror());
        return 2;
    }
    return serve(fd);
}
