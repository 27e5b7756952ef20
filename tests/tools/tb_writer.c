/* TEST INFRASTRUCTURE: an INDEPENDENT writer of TensorFlow "TensorBundle V2" checkpoints (<prefix>.index +
 * <prefix>.data-00000-of-00001), written from the format description only -- tensorflow/core/util/tensor_bundle (BundleHeaderProto,
 * BundleEntryProto), tensorflow/core/lib/io/{table_builder,block_builder,format}.cc (= the LevelDB table format) and
 * lib/hash/crc32c (Castagnoli, masked) -- in a different language and with none of the code of Data_utils/tf_checkpoint.py, whose READER
 * it cross-checks (tests/test_tf_checkpoint.py).  No TensorFlow exists in this environment, so this is not a TF-written file; it removes the
 * "reader and writer share one author's code path" weakness, not the "never met real TF output" one (INTEGRATION.md says so).
 *
 * It exercises what the Python writer's fixtures never produce: several data blocks (block size 1 KiB), prefix-compressed keys with restart
 * points every 16 entries, an index block with restart interval 1, proto3 default-field omission (shard_id / offset 0 are absent).
 *
 * usage: tb_writer <prefix> <ntensors>     tensor t: name "model/layer%03d/<weights|biases>", float32,
 *        shape [3,3,(t%5)+1,(t%7)+2] for weights / [(t%7)+2] for biases, element e = (float)(sin(0.37*t + 0.011*e)). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned char* p; size_t n, cap; } Buf;
static void put(Buf* b, const void* s, size_t n) {
    if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 64; b->p = (unsigned char*)realloc(b->p, b->cap); }
    memcpy(b->p + b->n, s, n); b->n += n;
}
static void put8(Buf* b, unsigned v) { unsigned char c = (unsigned char)v; put(b, &c, 1); }
static void put_fixed32(Buf* b, uint32_t v) { unsigned char c[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)}; put(b, c, 4); }
static void put_varint(Buf* b, uint64_t v) { while (v >= 128) { put8(b, (unsigned)(v & 127) | 128); v >>= 7; } put8(b, (unsigned)v); }

static uint32_t crc_table[256];
static void crc_init(void) {
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1; crc_table[i] = c; }
}
static uint32_t crc32c_ext(uint32_t crc, const unsigned char* d, size_t n) {
    uint32_t c = crc ^ 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ d[i]) & 255] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
static uint32_t crc_mask(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

/* ---- block builder (block_builder.cc): entries with prefix compression + restart array ---------------------------------- */
typedef struct { Buf data; uint32_t restarts[4096]; int nrestarts, counter, interval; char last[512]; int lastlen; } Block;
static void block_reset(Block* k, int interval) { k->data.n = 0; k->nrestarts = 1; k->restarts[0] = 0; k->counter = 0; k->interval = interval; k->lastlen = 0; }
static void block_add(Block* k, const char* key, int klen, const unsigned char* val, size_t vlen) {
    int shared = 0;
    if (k->counter < k->interval) { while (shared < klen && shared < k->lastlen && key[shared] == k->last[shared]) ++shared; }
    else { k->restarts[k->nrestarts++] = (uint32_t)k->data.n; k->counter = 0; }
    put_varint(&k->data, (uint64_t)shared); put_varint(&k->data, (uint64_t)(klen - shared)); put_varint(&k->data, vlen);
    put(&k->data, key + shared, (size_t)(klen - shared)); put(&k->data, val, vlen);
    memcpy(k->last, key, (size_t)klen); k->lastlen = klen; ++k->counter;
}
static void block_finish(Block* k) { for (int i = 0; i < k->nrestarts; ++i) put_fixed32(&k->data, k->restarts[i]); put_fixed32(&k->data, (uint32_t)k->nrestarts); }

/* writes block contents + 5-byte trailer (type 0 = uncompressed, masked crc32c over contents + type) ; returns (offset, size) */
static void write_block(FILE* f, Block* k, uint64_t* file_off, uint64_t* off, uint64_t* size) {
    block_finish(k);
    unsigned char type = 0;
    uint32_t crc = crc32c_ext(0, k->data.p, k->data.n);
    crc = crc_mask(crc32c_ext(crc, &type, 1));
    unsigned char tr[5] = {type, (unsigned char)crc, (unsigned char)(crc >> 8), (unsigned char)(crc >> 16), (unsigned char)(crc >> 24)};
    fwrite(k->data.p, 1, k->data.n, f); fwrite(tr, 1, 5, f);
    *off = *file_off; *size = k->data.n; *file_off += k->data.n + 5;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: tb_writer <prefix> <ntensors>\n"); return 2; }
    const int nt = atoi(argv[2]);
    crc_init();
    char path[1024];
    snprintf(path, sizeof path, "%s.data-00000-of-00001", argv[1]);
    FILE* fd = fopen(path, "wb");
    snprintf(path, sizeof path, "%s.index", argv[1]);
    FILE* fi = fopen(path, "wb");
    if (!fd || !fi) { perror("open"); return 1; }

    static Block data_block, index_block, meta_block;
    block_reset(&data_block, 16); block_reset(&index_block, 1); block_reset(&meta_block, 16);
    uint64_t file_off = 0, data_off = 0;
    const size_t BLOCK_SIZE = 1024;
    char pending_key[512]; int pending_len = -1; uint64_t pend_off = 0, pend_size = 0;

    /* entries must be added in key order: "" (header) first, then names sorted bytewise: layerNNN/biases < layerNNN/weights */
    Buf val = {0, 0, 0};
    /* BundleHeaderProto{ num_shards = 1 (field 1), endianness = LITTLE = 0 (omitted), version = VersionDef{ producer = 1 } (field 3) } */
    val.n = 0; put8(&val, (1 << 3) | 0); put_varint(&val, 1); put8(&val, (3 << 3) | 2); put_varint(&val, 2); put8(&val, (1 << 3) | 0); put_varint(&val, 1);
    block_add(&data_block, "", 0, val.p, val.n);

    for (int t = 0; t < nt; ++t)
        for (int which = 0; which < 2; ++which) {          /* 0 = biases, 1 = weights (sorted order) */
            char key[256];
            const int klen = snprintf(key, sizeof key, "model/layer%03d/%s", t, which ? "weights" : "biases");
            int dims[4], nd;
            if (which) { dims[0] = 3; dims[1] = 3; dims[2] = (t % 5) + 1; dims[3] = (t % 7) + 2; nd = 4; } else { dims[0] = (t % 7) + 2; nd = 1; }
            size_t count = 1;
            for (int d = 0; d < nd; ++d) count *= (size_t)dims[d];
            float* x = (float*)malloc(count * 4);
            for (size_t e = 0; e < count; ++e) x[e] = (float)sin(0.37 * t + 0.011 * (double)e + (which ? 0.0 : 1.0));
            fwrite(x, 4, count, fd);
            const uint32_t crc = crc_mask(crc32c_ext(0, (const unsigned char*)x, count * 4));
            free(x);
            /* BundleEntryProto{ dtype = DT_FLOAT = 1 (1), shape (2) = TensorShapeProto{ dim (2) = Dim{ size (1) } ... }, shard_id = 0 (omitted),
             *                   offset (4, omitted when 0), size (5), crc32c (6, fixed32) } */
            Buf shape = {0, 0, 0};
            for (int d = 0; d < nd; ++d) { Buf dim = {0, 0, 0}; put8(&dim, (1 << 3) | 0); put_varint(&dim, (uint64_t)dims[d]);
                                           put8(&shape, (2 << 3) | 2); put_varint(&shape, dim.n); put(&shape, dim.p, dim.n); free(dim.p); }
            val.n = 0;
            put8(&val, (1 << 3) | 0); put_varint(&val, 1);
            put8(&val, (2 << 3) | 2); put_varint(&val, shape.n); put(&val, shape.p, shape.n);
            if (data_off) { put8(&val, (4 << 3) | 0); put_varint(&val, data_off); }
            put8(&val, (5 << 3) | 0); put_varint(&val, count * 4);
            put8(&val, (6 << 3) | 5); put_fixed32(&val, crc);
            free(shape.p);
            data_off += count * 4;
            /* table_builder.cc: the index entry of a finished block is added when the NEXT key arrives (any separator >= last key works) */
            if (pending_len >= 0) {
                Buf h = {0, 0, 0}; put_varint(&h, pend_off); put_varint(&h, pend_size);
                block_add(&index_block, pending_key, pending_len, h.p, h.n); free(h.p); pending_len = -1;
            }
            block_add(&data_block, key, klen, val.p, val.n);
            if (data_block.data.n + 4 * (size_t)data_block.nrestarts + 4 >= BLOCK_SIZE) {
                memcpy(pending_key, key, (size_t)klen); pending_len = klen;
                write_block(fi, &data_block, &file_off, &pend_off, &pend_size);
                block_reset(&data_block, 16);
            }
        }
    if (data_block.data.n > 0) {
        if (pending_len >= 0) { Buf h = {0, 0, 0}; put_varint(&h, pend_off); put_varint(&h, pend_size); block_add(&index_block, pending_key, pending_len, h.p, h.n); free(h.p); }
        memcpy(pending_key, data_block.last, (size_t)data_block.lastlen); pending_len = data_block.lastlen;
        write_block(fi, &data_block, &file_off, &pend_off, &pend_size);
    }
    if (pending_len >= 0) { Buf h = {0, 0, 0}; put_varint(&h, pend_off); put_varint(&h, pend_size); block_add(&index_block, pending_key, pending_len, h.p, h.n); free(h.p); }
    uint64_t meta_off, meta_size, idx_off, idx_size;
    write_block(fi, &meta_block, &file_off, &meta_off, &meta_size);
    write_block(fi, &index_block, &file_off, &idx_off, &idx_size);
    /* footer (format.cc): metaindex handle, index handle, zero padding to 40 bytes, magic 0xdb4775248b80fb57 as two little-endian fixed32 */
    Buf foot = {0, 0, 0};
    put_varint(&foot, meta_off); put_varint(&foot, meta_size); put_varint(&foot, idx_off); put_varint(&foot, idx_size);
    while (foot.n < 40) put8(&foot, 0);
    put_fixed32(&foot, 0x8b80fb57u); put_fixed32(&foot, 0xdb477524u);
    fwrite(foot.p, 1, foot.n, fi);
    fclose(fi); fclose(fd);
    return 0;
}
