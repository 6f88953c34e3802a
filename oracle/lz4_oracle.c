/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the LZ4 frame codec the reference's capture path uses.
 *
 * Nothing in the shipped product may include, link or call this file.
 *
 * The reference (src/compressor.rs:98,126-129,138-144; src/decompressor.rs:198,216-235) calls the
 * third-party crate `lzzzz` ("*", unpinned; it vendors liblz4), which is absent from /root/reference.
 * This file restates the published LZ4 algorithms from the lz4 1.9.x sources:
 *   - LZ4F frame format (magic, FLG/BD/HC header, block headers, end mark), lz4frame.c
 *   - LZ4F_compressUpdate with autoFlush=1, blockLinked, 64 KiB blocks, no checksums, level 0
 *     (the preferences the reference builds at src/compressor.rs:138-144)
 *   - LZ4_compress_fast_continue / LZ4_compress_generic (byU32 table, 64-bit hash5), lz4.c
 *   - LZ4_saveDict, LZ4_renormDictT
 * Pinned in tests against liblz4.so.1 (1.9.3) shipped in the image: byte-equal streams for the same write sequence.
 * Every lz4o_update() call corresponds to ONE Write::write_all() in the reference (= one LZ4F_compressUpdate).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KB *(1 << 10)
#define MINMATCH 4
#define LASTLITERALS 5
#define MFLIMIT 12
#define LZ4_MINLENGTH (MFLIMIT + 1)
#define HASHLOG 12
#define HASH_SIZE_U32 (1 << HASHLOG)
#define SKIP_TRIGGER 6
#define DISTANCE_MAX 65535
#define ML_BITS 4
#define ML_MASK 15u
#define RUN_MASK 15u
#define BLOCK_SIZE (64 KB)
#define BLOCK_UNCOMPRESSED_FLAG 0x80000000u

typedef struct {
    uint32_t hash[HASH_SIZE_U32];
    uint32_t current_offset;
    const uint8_t* dictionary;
    uint32_t dict_size;
} lz4_state;

typedef struct lz4o_stream {
    lz4_state st;
    uint8_t* tmp_buff;          /* 64 KiB dictionary save area (LZ4F tmpBuff) */
} lz4o_stream;

static uint32_t read32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t read64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static void write_le32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* LZ4_hash5 on a little-endian 64-bit target, byU32 table */
static uint32_t hash5(const uint8_t* p) {
    const uint64_t prime5bytes = 889523592379ULL;
    return (uint32_t)(((read64(p) << 24) * prime5bytes) >> (64 - HASHLOG));
}

static unsigned count_match(const uint8_t* ip, const uint8_t* match, const uint8_t* limit) {
    const uint8_t* const start = ip;
    while (ip < limit && *ip == *match) { ip++; match++; }
    return (unsigned)(ip - start);
}

enum { WITH_PREFIX, USING_EXT_DICT };

/* LZ4_compress_generic_validated, limitedOutput, byU32.  Returns 0 when the output does not fit. */
static int compress_generic(lz4_state* cctx, const uint8_t* source, uint8_t* dest, int input_size, int max_output,
                            int dict_directive, int dict_small) {
    const uint8_t* ip = source;
    const uint32_t start_index = cctx->current_offset;
    const uint8_t* base = source - start_index;
    const uint8_t* low_limit;
    const uint8_t* const dictionary = cctx->dictionary;
    const uint32_t dict_size = cctx->dict_size;
    const int maybe_ext = dict_directive == USING_EXT_DICT;
    const uint32_t prefix_idx_limit = start_index - dict_size;
    const uint8_t* const dict_end = dictionary ? dictionary + dict_size : dictionary;
    const uint8_t* anchor = source;
    const uint8_t* const iend = ip + input_size;
    const uint8_t* const mflimit_plus_one = iend - MFLIMIT + 1;
    const uint8_t* const matchlimit = iend - LASTLITERALS;
    const uint8_t* dict_base = !dictionary ? NULL : dictionary + dict_size - start_index;
    uint8_t* op = dest;
    uint8_t* const olimit = op + max_output;
    uint32_t offset = 0, forward_h;
    const uint8_t* match;
    uint8_t* token;

    low_limit = source - (dict_directive == WITH_PREFIX ? dict_size : 0);
    cctx->dict_size += (uint32_t)input_size;
    cctx->current_offset += (uint32_t)input_size;

    if (input_size < LZ4_MINLENGTH) goto last_literals;

    cctx->hash[hash5(ip)] = (uint32_t)(ip - base);
    ip++; forward_h = hash5(ip);

    for (;;) {
        {   const uint8_t* forward_ip = ip;
            int step = 1;
            int search_match_nb = 1 << SKIP_TRIGGER;        /* acceleration 1 */
            for (;;) {
                uint32_t const h = forward_h;
                uint32_t const current = (uint32_t)(forward_ip - base);
                uint32_t match_index = cctx->hash[h];
                ip = forward_ip;
                forward_ip += step;
                step = (search_match_nb++ >> SKIP_TRIGGER);
                if (forward_ip > mflimit_plus_one) goto last_literals;
                if (dict_directive == USING_EXT_DICT) {
                    if (match_index < start_index) { match = dict_base + match_index; low_limit = dictionary; }
                    else { match = base + match_index; low_limit = source; }
                } else {
                    match = base + match_index;
                }
                forward_h = hash5(forward_ip);
                cctx->hash[h] = current;
                if (dict_small && match_index < prefix_idx_limit) continue;
                if (match_index + DISTANCE_MAX < current) continue;
                if (read32(match) == read32(ip)) { if (maybe_ext) offset = current - match_index; break; }
            }
        }
        while ((ip > anchor) && (match > low_limit) && (ip[-1] == match[-1])) { ip--; match--; }
        {   unsigned const lit_length = (unsigned)(ip - anchor);
            token = op++;
            if (op + lit_length + (2 + 1 + LASTLITERALS) + (lit_length / 255) > olimit) return 0;
            if (lit_length >= RUN_MASK) {
                int len = (int)(lit_length - RUN_MASK);
                *token = (uint8_t)(RUN_MASK << ML_BITS);
                for (; len >= 255; len -= 255) *op++ = 255;
                *op++ = (uint8_t)len;
            } else *token = (uint8_t)(lit_length << ML_BITS);
            memcpy(op, anchor, lit_length);
            op += lit_length;
        }
next_match:
        if (maybe_ext) { op[0] = (uint8_t)offset; op[1] = (uint8_t)(offset >> 8); op += 2; }
        else { uint32_t o = (uint32_t)(ip - match); op[0] = (uint8_t)o; op[1] = (uint8_t)(o >> 8); op += 2; }
        {   unsigned match_code;
            if (dict_directive == USING_EXT_DICT && low_limit == dictionary) {
                const uint8_t* limit = ip + (dict_end - match);
                if (limit > matchlimit) limit = matchlimit;
                match_code = count_match(ip + MINMATCH, match + MINMATCH, limit);
                ip += (size_t)match_code + MINMATCH;
                if (ip == limit) {
                    unsigned const more = count_match(limit, source, matchlimit);
                    match_code += more; ip += more;
                }
            } else {
                match_code = count_match(ip + MINMATCH, match + MINMATCH, matchlimit);
                ip += (size_t)match_code + MINMATCH;
            }
            if (op + (1 + LASTLITERALS) + (match_code + 240) / 255 > olimit) return 0;
            if (match_code >= ML_MASK) {
                *token += ML_MASK;
                match_code -= ML_MASK;
                while (match_code >= 255) { *op++ = 255; match_code -= 255; }
                *op++ = (uint8_t)match_code;
            } else *token += (uint8_t)match_code;
        }
        anchor = ip;
        if (ip >= mflimit_plus_one) break;
        cctx->hash[hash5(ip - 2)] = (uint32_t)(ip - 2 - base);
        {   uint32_t const h = hash5(ip);
            uint32_t const current = (uint32_t)(ip - base);
            uint32_t match_index = cctx->hash[h];
            if (dict_directive == USING_EXT_DICT) {
                if (match_index < start_index) { match = dict_base + match_index; low_limit = dictionary; }
                else { match = base + match_index; low_limit = source; }
            } else match = base + match_index;
            cctx->hash[h] = current;
            if ((dict_small ? (match_index >= prefix_idx_limit) : 1)
                && (match_index + DISTANCE_MAX >= current)
                && (read32(match) == read32(ip))) {
                token = op++; *token = 0;
                if (maybe_ext) offset = current - match_index;
                goto next_match;
            }
        }
        forward_h = hash5(++ip);
    }
last_literals:
    {   size_t last_run = (size_t)(iend - anchor);
        if (op + last_run + 1 + ((last_run + 255 - RUN_MASK) / 255) > olimit) return 0;
        if (last_run >= RUN_MASK) {
            size_t acc = last_run - RUN_MASK;
            *op++ = (uint8_t)(RUN_MASK << ML_BITS);
            for (; acc >= 255; acc -= 255) *op++ = 255;
            *op++ = (uint8_t)acc;
        } else *op++ = (uint8_t)(last_run << ML_BITS);
        memcpy(op, anchor, last_run);
        op += last_run;
    }
    return (int)(op - dest);
}

static void renorm_dict(lz4_state* d, int next_size) {
    if (d->current_offset + (uint32_t)next_size > 0x80000000u) {
        uint32_t const delta = d->current_offset - 64 KB;
        const uint8_t* dict_end = d->dictionary + d->dict_size;
        for (int i = 0; i < HASH_SIZE_U32; i++) {
            if (d->hash[i] < delta) d->hash[i] = 0; else d->hash[i] -= delta;
        }
        d->current_offset = 64 KB;
        if (d->dict_size > 64 KB) d->dict_size = 64 KB;
        d->dictionary = dict_end - d->dict_size;
    }
}

/* LZ4_compress_fast_continue, acceleration 1, no dictCtx */
static int compress_fast_continue(lz4_state* s, const uint8_t* source, uint8_t* dest, int input_size, int max_output) {
    const uint8_t* dict_end = s->dictionary + s->dict_size;
    renorm_dict(s, input_size);
    dict_end = s->dictionary + s->dict_size;
    if ((s->dict_size - 1 < 4 - 1) && (dict_end != source)) {   /* intentional underflow */
        s->dict_size = 0; s->dictionary = source; dict_end = source;
    }
    {   const uint8_t* source_end = source + input_size;
        if ((source_end > s->dictionary) && (source_end < dict_end)) {
            s->dict_size = (uint32_t)(dict_end - source_end);
            if (s->dict_size > 64 KB) s->dict_size = 64 KB;
            if (s->dict_size < 4) s->dict_size = 0;
            s->dictionary = dict_end - s->dict_size;
        }
    }
    if (dict_end == source) {
        int small = (s->dict_size < 64 KB) && (s->dict_size < s->current_offset);
        return compress_generic(s, source, dest, input_size, max_output, WITH_PREFIX, small);
    }
    {   int small = (s->dict_size < 64 KB) && (s->dict_size < s->current_offset);
        int r = compress_generic(s, source, dest, input_size, max_output, USING_EXT_DICT, small);
        s->dictionary = source;
        s->dict_size = (uint32_t)input_size;
        return r;
    }
}

static int save_dict(lz4_state* d, uint8_t* safe, int dict_size) {
    const uint8_t* const prev_end = d->dictionary + d->dict_size;
    if ((uint32_t)dict_size > 64 KB) dict_size = 64 KB;
    if ((uint32_t)dict_size > d->dict_size) dict_size = (int)d->dict_size;
    if (dict_size > 0) memmove(safe, prev_end - dict_size, (size_t)dict_size);
    d->dictionary = safe;
    d->dict_size = (uint32_t)dict_size;
    return dict_size;
}

/* LZ4F_makeBlock without block checksum */
static size_t make_block(lz4o_stream* s, uint8_t* dst, const uint8_t* src, size_t src_size) {
    uint32_t c = (uint32_t)compress_fast_continue(&s->st, src, dst + 4, (int)src_size, (int)src_size - 1);
    if (c == 0) {
        c = (uint32_t)src_size;
        write_le32(dst, c | BLOCK_UNCOMPRESSED_FLAG);
        memcpy(dst + 4, src, src_size);
    } else write_le32(dst, c);
    return 4 + c;
}

/* XXH32 (seed 0) -- only needed for the 2-byte frame descriptor checksum */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t xxh32_small(const uint8_t* p, size_t len) {
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    uint32_t h = P5 + (uint32_t)len;     /* len < 16 */
    while (len >= 4) { h += read32(p) * P3; h = rotl32(h, 17) * P4; p += 4; len -= 4; }
    while (len > 0) { h += (*p) * P5; h = rotl32(h, 11) * P1; p++; len--; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

lz4o_stream* lz4o_create(void) {
    lz4o_stream* s = (lz4o_stream*)calloc(1, sizeof *s);
    if (!s) return NULL;
    /* 8 guard bytes on each side: hash5 reads 8 bytes and count_match never passes the segment end */
    s->tmp_buff = (uint8_t*)calloc(1, 64 KB + 64);
    if (!s->tmp_buff) { free(s); return NULL; }
    return s;
}

void lz4o_destroy(lz4o_stream* s) { if (s) { free(s->tmp_buff); free(s); } }

/* LZ4F_compressBound for these preferences: every block may be stored, plus header and end mark */
size_t lz4o_bound(size_t n) { return n + 4 * (n / BLOCK_SIZE + 1) + 7 + 4; }

/* LZ4F_compressBegin: magic, FLG (version 01, block-linked), BD (64 KiB), HC */
size_t lz4o_begin(lz4o_stream* s, uint8_t* dst, size_t cap) {
    if (cap < 7) return 0;
    memset(&s->st, 0, sizeof s->st);
    write_le32(dst, 0x184D2204u);
    dst[4] = 0x40; dst[5] = 0x40;
    dst[6] = (uint8_t)(xxh32_small(dst + 4, 2) >> 8);
    return 7;
}

/* LZ4F_compressUpdate(src, n) with autoFlush: full 64 KiB blocks, then the remainder, then LZ4F_localSaveDict */
size_t lz4o_update(lz4o_stream* s, uint8_t* dst, size_t cap, const uint8_t* src, size_t n) {
    uint8_t* op = dst;
    const uint8_t* ip = src; const uint8_t* const end = src + n;
    if (n == 0) return 0;
    if (cap < lz4o_bound(n) - 11) return 0;
    while ((size_t)(end - ip) >= BLOCK_SIZE) { op += make_block(s, op, ip, BLOCK_SIZE); ip += BLOCK_SIZE; }
    if (ip < end) { op += make_block(s, op, ip, (size_t)(end - ip)); ip = end; }
    save_dict(&s->st, s->tmp_buff, 64 KB);
    return (size_t)(op - dst);
}

/* LZ4F_compressEnd: nothing buffered (autoFlush), so just the end mark */
size_t lz4o_end(lz4o_stream* s, uint8_t* dst, size_t cap) {
    (void)s;
    if (cap < 4) return 0;
    write_le32(dst, 0);
    return 4;
}

/* ---- decoder -------------------------------------------------------------------------------------------------- */
/* One LZ4 block, history = everything already in dst[0..pos).  Returns new pos or -1 on malformed input. */
static int64_t decode_block(const uint8_t* src, size_t n, uint8_t* dst, size_t pos, size_t cap) {
    const uint8_t* ip = src; const uint8_t* const iend = src + n;
    while (ip < iend) {
        unsigned token = *ip++;
        size_t lit = token >> ML_BITS;
        if (lit == RUN_MASK) { unsigned b; do { if (ip >= iend) return -1; b = *ip++; lit += b; } while (b == 255); }
        if ((size_t)(iend - ip) < lit || cap - pos < lit) return -1;
        memcpy(dst + pos, ip, lit); pos += lit; ip += lit;
        if (ip >= iend) break;                      /* last sequence: literals only */
        if (iend - ip < 2) return -1;
        size_t off = (size_t)ip[0] | ((size_t)ip[1] << 8); ip += 2;
        size_t ml = token & ML_MASK;
        if (ml == ML_MASK) { unsigned b; do { if (ip >= iend) return -1; b = *ip++; ml += b; } while (b == 255); }
        ml += MINMATCH;
        if (off == 0 || off > pos || cap - pos < ml) return -1;
        for (size_t i = 0; i < ml; i++) dst[pos + i] = dst[pos + i - off];
        pos += ml;
    }
    return (int64_t)pos;
}

/* Decodes one LZ4 frame (block-linked or independent, no checksums beyond the flags parsed here).
 * A truncated tail is tolerated: decoding stops at the last complete block (reference src/decompressor.rs:216-235
 * tolerates an unclean end).  *consumed = bytes of src used, *complete = 1 when the end mark was seen.
 * Returns bytes produced, or -1 on a malformed header/block. */
int64_t lz4o_decompress_frame(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* consumed, int* complete) {
    size_t ip = 0, pos = 0;
    if (complete) *complete = 0;
    if (consumed) *consumed = 0;
    if (n < 7 || read32(src) != 0x184D2204u) return -1;
    uint8_t flg = src[4];
    if ((flg >> 6) != 1) return -1;
    int block_checksum = (flg >> 4) & 1, content_size = (flg >> 3) & 1, content_checksum = (flg >> 2) & 1, dict_id = flg & 1;
    ip = 6 + (content_size ? 8 : 0) + (dict_id ? 4 : 0) + 1;
    if (n < ip) return -1;
    for (;;) {
        if (n - ip < 4) break;
        uint32_t bh = read32(src + ip);
        if (bh == 0) { ip += 4; if (content_checksum) { if (n - ip < 4) break; ip += 4; } if (complete) *complete = 1; break; }
        size_t bsz = bh & 0x7FFFFFFFu;
        if (n - ip - 4 < bsz + (block_checksum ? 4 : 0)) break;     /* truncated block */
        if (bh & BLOCK_UNCOMPRESSED_FLAG) {
            if (cap - pos < bsz) return -1;
            memcpy(dst + pos, src + ip + 4, bsz); pos += bsz;
        } else {
            int64_t r = decode_block(src + ip + 4, bsz, dst, pos, cap);
            if (r < 0) return -1;
            pos = (size_t)r;
        }
        ip += 4 + bsz + (block_checksum ? 4 : 0);
        if (consumed) *consumed = ip;
    }
    if (consumed) *consumed = ip;
    return (int64_t)pos;
}

int lz4o_abi_version(void) { return 1; }
