// TEST INFRASTRUCTURE ONLY (see oracle/README.md): where the trace points of oracle/trace_hooks.h write.  One file (HAVOC_TRACE_FILE), 64-byte
// records { uint32 thread, uint16 kind, uint16 n, int32 value[14] } in the order the calls were made; a record carries the index of the thread
// that made it, so the reader can follow each thread's searches separately when the encoder runs several.  Nothing is written when the
// variable is not set.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

struct Sink
{
    FILE *f = nullptr;
    std::mutex m;
    std::atomic<uint32_t> threads{0};
    Sink()
    {
        const char *path = getenv("HAVOC_TRACE_FILE");
        if (path && *path)
        {
            f = fopen(path, "wb");
            if (!f)
            {
                fprintf(stderr, "HAVOC_TRACE_FILE: cannot open %s\n", path);
                abort();
            }
            setvbuf(f, nullptr, _IOFBF, 1 << 20);
        }
    }
    ~Sink()
    {
        if (f) fclose(f);
    }
};

Sink &sink()
{
    static Sink s;
    return s;
}

} // namespace

extern "C" void havoc_trace_emit(int kind, int n, const int32_t *values)
{
    Sink &s = sink();
    if (!s.f) return;
    static thread_local uint32_t me = s.threads.fetch_add(1);
    struct
    {
        uint32_t thread;
        uint16_t kind, n;
        int32_t value[14];
    } r;
    memset(&r, 0, sizeof(r));
    r.thread = me;
    r.kind = uint16_t(kind);
    r.n = uint16_t(n);
    memcpy(r.value, values, size_t(n > 14 ? 14 : n) * 4);
    std::lock_guard<std::mutex> lock(s.m);
    fwrite(&r, sizeof(r), 1, s.f);
}
