// Debugging aid: print the native call stack when the process dies of SIGSEGV / SIGBUS / SIGABRT (uspmv_debug_backtrace_on_crash,
// USPMV_BACKTRACE=1 in the harness and the Python front-end).  Frames are module + offset (backtrace_symbols_fd), enough to
// name the shared library and function a crash inside a runtime call sits in.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <cstring>

#include "uspmv_internal.hpp"

namespace {
void on_crash(int sig) {
    const char *msg = "[uspmv] fatal signal, native call stack:\n";
    if (write(2, msg, strlen(msg)) < 0) {}
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
}  // namespace

extern "C" int uspmv_debug_backtrace_on_crash(int on) {
    void *warm[2];
    (void)backtrace(warm, 2);   // loads libgcc's unwinder now, not inside the handler
    for (int sig : {SIGSEGV, SIGBUS, SIGABRT}) signal(sig, on ? on_crash : SIG_DFL);
    return USPMV_OK;
}
