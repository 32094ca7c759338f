/* TEST INFRASTRUCTURE ONLY (see oracle/README.md): plain-C restatement of the reference's sample-adaptive-offset primitives --
 * the two filters of /root/reference/turing/sao.cpp:33-92 and the statistics functions of turing/EncSao.h:62-283 the encoder's SAO
 * decision reads.  Pinned against those sources compiled into oracle/_ref (oracle/ref_shim_sao.cpp) by tests/test_sao.py. */
#include "havoc_oracle.h"

static int sao_get(const void *p, long i, int S) { return S == 1 ? ((const uint8_t *)p)[i] : ((const uint16_t *)p)[i]; }
static int sao_sign(int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }
static const int kCategory[5] = { 1, 2, 0, 3, 4 };      /* EncSao.h:157: 2 + sign + sign -> category */

/* out[0..39]: for each edge class c = 0..3: E[5] at 10c, count[5] at 10c + 5 (EncSao.h:151-283); out[40..71] band E[32],
 * out[72..103] band count[32], out[104] the band position band_offset_luma_stats returns (EncSao.h:111-148).  The statistics
 * cover the block WITHOUT its outermost ring of samples; shift = bitDepth - 8. */
void oracle_sao_stats(const void *src, intptr_t ss, const void *rec, intptr_t rs, int w, int h, int shift, int S, int64_t *out)
{
    static const int dx[4][2] = { { -1, 1 }, { 0, 0 }, { -1, 1 }, { 1, -1 } }, dy[4][2] = { { 0, 0 }, { -1, 1 }, { -1, 1 }, { -1, 1 } };
    for (int k = 0; k < 105; ++k) out[k] = 0;
    for (int y = 1; y < h - 1; ++y)
    {
        for (int x = 1; x < w - 1; ++x)
        {
            const int c = sao_get(rec, y * rs + x, S), diff = sao_get(src, y * ss + x, S) - c;
            for (int cls = 0; cls < 4; ++cls)
            {
                const int a = sao_sign(c - sao_get(rec, (y + dy[cls][0]) * rs + x + dx[cls][0], S));
                const int b = sao_sign(c - sao_get(rec, (y + dy[cls][1]) * rs + x + dx[cls][1], S));
                const int cat = kCategory[2 + a + b];
                out[10 * cls + cat] += diff;
                out[10 * cls + 5 + cat]++;
            }
            out[40 + (c >> (3 + shift))] += diff;
            out[72 + (c >> (3 + shift))]++;
        }
        /* EncSao.h:166-171 then :174-176 with x = 1: the horizontal class takes the first interior sample of a row twice, the second
         * time with its two signs cancelling (category 0, which no offset uses) */
        if (w > 2)
        {
            out[0 + kCategory[2]] += sao_get(src, y * ss + 1, S) - sao_get(rec, y * rs + 1, S);
            out[5 + kCategory[2]]++;
        }
    }
    {   /* EncSao.h:134-147: the four consecutive bands holding most samples */
        int64_t best = 0;
        int start = 0;
        for (int b = 0; b < 29; ++b)
        {
            const int64_t cum = out[72 + b] + out[73 + b] + out[74 + b] + out[75 + b];
            if (cum > best)
            {
                best = cum;
                start = b;
            }
        }
        out[104] = start + 1 < 2 ? 2 : start + 1;
    }
}

/* sao.cpp:33-92.  type 1: band offset, `offsets` = the 32-entry table; type 2: edge offset, `offsets` = SaoOffsetVal[5]; type 0: copy */
void oracle_sao_filter(void *dst, intptr_t ds, const void *src, intptr_t ss, int w, int h, int type, int eoClass, const int16_t *offsets, int bitDepth, int S)
{
    static const int hLookup[4] = { -1, 0, -1, 1 }, vLookup[4] = { 0, -1, -1, -1 };
    const int mx = (1 << bitDepth) - 1;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
        {
            const int c = sao_get(src, y * ss + x, S);
            int v = c;
            if (type == 1)
                v = c + offsets[c >> (bitDepth - 5)];
            else if (type == 2)
            {
                int idx = 2 + sao_sign(c - sao_get(src, (y + vLookup[eoClass]) * ss + x + hLookup[eoClass], S))
                            + sao_sign(c - sao_get(src, (y - vLookup[eoClass]) * ss + x - hLookup[eoClass], S));
                if (idx <= 2) idx = idx == 2 ? 0 : idx + 1;
                v = c + offsets[idx];
            }
            v = v < 0 ? 0 : (v > mx ? mx : v);
            if (S == 1) ((uint8_t *)dst)[y * ds + x] = (uint8_t)v;
            else ((uint16_t *)dst)[y * ds + x] = (uint16_t)v;
        }
}


/* band_offset_chroma_stats, turing/EncSao.h:62-109: one band histogram over the interiors of the Cb and Cr blocks; out[0..31] E, out[32..63]
 * count, out[64] the band position returned */
void oracle_sao_band_chroma(const void *src_u, const void *src_v, intptr_t ss, const void *rec_u, const void *rec_v, intptr_t rs, int w, int h, int shift, int S,
                            int64_t *out)
{
    for (int k = 0; k < 65; ++k) out[k] = 0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x)
            for (int p = 0; p < 2; ++p)
            {
                const int c = sao_get(p ? rec_v : rec_u, y * rs + x, S);
                out[c >> (3 + shift)] += sao_get(p ? src_v : src_u, y * ss + x, S) - c;
                out[32 + (c >> (3 + shift))]++;
            }
    int64_t best = 0;
    int start = 0;
    for (int b = 0; b < 29; ++b)
    {
        const int64_t cum = out[32 + b] + out[33 + b] + out[34 + b] + out[35 + b];
        if (cum > best)
        {
            best = cum;
            start = b;
        }
    }
    out[64] = start + 1 < 2 ? 2 : start + 1;
}
