// TEST INFRASTRUCTURE ONLY (see oracle/README.md): where the trace points of oracle/trace_hooks.h write.  One file (HAVOC_TRACE_FILE), 64-byte
// records { uint32 thread, uint16 kind, uint16 n, int32 value[14] } in the order the calls were made; a record carries the index of the thread
// that made it, so the reader can follow each thread's searches separately when the encoder runs several.  Nothing is written when the
// variable is not set.  HAVOC_TRACE_SUMMARY=<file> (with or without a trace file): only counts -- records per kind, searches by block size,
// intra partitions by size -- written as JSON at exit: how profiles/measure_call_mix.py counts the searches of a 1080p / 4K encode.
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
    const char *summary = nullptr;
    std::atomic<long> kinds[32], uni[17][17], bi[17][17], intra[8];
    Sink()
    {
        for (auto &k : kinds) k = 0;
        for (auto &r : uni) for (auto &c : r) c = 0;
        for (auto &r : bi) for (auto &c : r) c = 0;
        for (auto &c : intra) c = 0;
        summary = getenv("HAVOC_TRACE_SUMMARY");
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
        if (summary && *summary)
            if (FILE *o = fopen(summary, "w"))
            {
                fprintf(o, "{\"records_by_kind\": {");
                for (int k = 0, first = 1; k < 32; ++k)
                    if (kinds[k].load())
                    {
                        fprintf(o, "%s\"%d\": %ld", first ? "" : ", ", k, kinds[k].load());
                        first = 0;
                    }
                fprintf(o, "}");
                const char *names[2] = {"uni_searches_by_size", "bi_searches_by_size"};
                for (int t = 0; t < 2; ++t)
                {
                    fprintf(o, ", \"%s\": {", names[t]);
                    for (int w = 1, first = 1; w <= 16; ++w)
                        for (int h = 1; h <= 16; ++h)
                            if (long n = (t ? bi : uni)[w][h].load())
                            {
                                fprintf(o, "%s\"%dx%d\": %ld", first ? "" : ", ", 4 * w, 4 * h, n);
                                first = 0;
                            }
                    fprintf(o, "}");
                }
                fprintf(o, ", \"intra_partitions_by_log2_size\": {");
                for (int l = 0, first = 1; l < 8; ++l)
                    if (intra[l].load())
                    {
                        fprintf(o, "%s\"%d\": %ld", first ? "" : ", ", l, intra[l].load());
                        first = 0;
                    }
                fprintf(o, "}}\n");
                fclose(o);
            }
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
    if (s.summary)
    {
        if (kind >= 0 && kind < 32) ++s.kinds[kind];
        if ((kind == 1 || kind == 9) && n >= 7 && values[5] >= 4 && values[5] <= 64 && values[6] >= 4 && values[6] <= 64)
            ++(kind == 1 ? s.uni : s.bi)[values[5] / 4][values[6] / 4];
        if (kind == 12 && n >= 4 && values[3] >= 0 && values[3] < 8) ++s.intra[values[3]];
    }
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
