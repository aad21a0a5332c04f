/* Test infrastructure: the C stack of whichever thread calls abort(), on stderr, before the process dies.
 * One full run of the -m gpu suite in a dozen ends with SIGABRT somewhere under a library call, and
 * Python's faulthandler only shows the Python frames of the thread that holds the interpreter.  tests/conftest.py compiles
 * this on first use (gcc -shared) and loads it with ctypes. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig)
{
	static const char head[] = "\n==== SIGABRT: C stack of the aborting thread (tests/csrc/abort_trace.c) ====\n";
	void *frames[64];
	ssize_t w = write(2, head, sizeof(head) - 1);
	(void) w;
	backtrace_symbols_fd(frames, backtrace(frames, 64), 2);
	signal(sig, SIG_DFL);
	raise(sig);
}

__attribute__((constructor)) static void install(void)
{
	void *warm[4];
	backtrace(warm, 4);                 /* loads libgcc now: not from inside the handler */
	struct sigaction sa;
	memset(&sa, 0, sizeof(sa));
	sa.sa_handler = on_abort;
	sigemptyset(&sa.sa_mask);
	sa.sa_flags = SA_NODEFER;
	sigaction(SIGABRT, &sa, NULL);
}
