// ParquetWriter (reference src/output/parquet.rs:23-300) without arrow / parquet crates: a minimal Apache Parquet
// file writer for the one table shape the reference emits.
//
//   schema  (parquet.rs:141-171)  start_s f32, end_s f32, scientific_name utf8, common_name utf8, confidence f32,
//                                 file utf8 (all required) + the optional metadata columns lat / lon f64, week u8,
//                                 model utf8, overlap / sensitivity / min_conf f32, species_list utf8 (nullable);
//                                 unknown column names are skipped
//   file    (parquet.rs:219-231)  the FILE NAME of the detection's path, the whole path when it has none
//   props   (parquet.rs:44-47)    writer version 2 (DataPageV2), SNAPPY
//
// One row group, one PLAIN-encoded DataPageV2 per column.  The snappy streams are literal-only (a valid snappy
// encoding; no match search): the reference's property is kept, the table a reader sees is the same.  The bytes differ
// from arrow-rs's (dictionary pages, statistics, the ARROW:schema key) but every reader sees the same schema and rows;
// tests/test_abi_and_host.py reads the file back with pyarrow.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host_internal.hpp"

namespace bhh {
namespace {

// ---- thrift compact protocol --------------------------------------------------------------------------
enum TType { T_TRUE = 1, T_FALSE = 2, T_BYTE = 3, T_I16 = 4, T_I32 = 5, T_I64 = 6, T_DOUBLE = 7, T_BINARY = 8, T_LIST = 9, T_STRUCT = 12 };
struct Thrift {
    std::string b;
    std::vector<int> last{0};
    void varint(uint64_t v) { while (v >= 0x80) { b.push_back((char)(v | 0x80)); v >>= 7; } b.push_back((char)v); }
    void zz(int64_t v) { varint(((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
    void field(int id, int type) {
        const int delta = id - last.back();
        if (delta > 0 && delta <= 15) b.push_back((char)((delta << 4) | type));
        else { b.push_back((char)type); zz(id); }
        last.back() = id;
    }
    void i32(int id, int32_t v) { field(id, T_I32); zz(v); }
    void i64(int id, int64_t v) { field(id, T_I64); zz(v); }
    void i8(int id, int8_t v) { field(id, T_BYTE); b.push_back((char)v); }
    void boolean(int id, bool v) { field(id, v ? T_TRUE : T_FALSE); }
    void str(int id, const std::string &s) { field(id, T_BINARY); varint(s.size()); b += s; }
    void begin_struct(int id) { field(id, T_STRUCT); last.push_back(0); }
    void begin_struct_elem() { last.push_back(0); }
    void end_struct() { b.push_back(0); last.pop_back(); }
    void list(int id, int elem_type, size_t n) {
        field(id, T_LIST);
        if (n < 15) b.push_back((char)((n << 4) | elem_type));
        else { b.push_back((char)(0xF0 | elem_type)); varint(n); }
    }
};

enum PType { P_INT32 = 1, P_FLOAT = 4, P_DOUBLE = 5, P_BYTE_ARRAY = 6 };
struct Column {
    std::string name;
    PType type;
    bool optional;
    int converted = -1;      // ConvertedType: UTF8 0, UINT_8 11
    int logical = 0;         // 1 = STRING, 10 = INTEGER(8, unsigned)
    std::string values;      // PLAIN-encoded non-null values
    size_t nulls = 0;
};

void put_f32(std::string &s, float v) { char b[4]; memcpy(b, &v, 4); s.append(b, 4); }
void put_str(std::string &s, const std::string &v) { uint32_t n = (uint32_t)v.size(); char b[4]; memcpy(b, &n, 4); s.append(b, 4); s += v; }

// snappy framing-less block: varint(uncompressed length) + literal elements of up to 2^16 bytes
std::string snappy_literal(const std::string &raw) {
    std::string o;
    uint64_t n = raw.size();
    while (n >= 0x80) { o.push_back((char)(n | 0x80)); n >>= 7; }
    o.push_back((char)n);
    for (size_t p = 0; p < raw.size();) {
        const size_t len = std::min<size_t>(raw.size() - p, 65536);
        if (len <= 60) o.push_back((char)((len - 1) << 2));
        else if (len <= 256) { o.push_back((char)(60 << 2)); o.push_back((char)(len - 1)); }
        else { o.push_back((char)(61 << 2)); o.push_back((char)((len - 1) & 0xFF)); o.push_back((char)((len - 1) >> 8)); }
        o.append(raw, p, len);
        p += len;
    }
    return o;
}

}  // namespace

int write_parquet_file(const std::string &path, const std::vector<Detection> &dets, const std::vector<std::string> &extra, std::string &err) {
    std::vector<Column> cols = {
        {"start_s", P_FLOAT, false}, {"end_s", P_FLOAT, false}, {"scientific_name", P_BYTE_ARRAY, false, 0, 1},
        {"common_name", P_BYTE_ARRAY, false, 0, 1}, {"confidence", P_FLOAT, false}, {"file", P_BYTE_ARRAY, false, 0, 1}};
    for (const auto &c : extra) {   // build_schema, parquet.rs:151-168
        if (c == "lat" || c == "lon") cols.push_back({c, P_DOUBLE, true});
        else if (c == "week") cols.push_back({c, P_INT32, true, 11, 10});
        else if (c == "model" || c == "species_list") cols.push_back({c, P_BYTE_ARRAY, true, 0, 1});
        else if (c == "overlap" || c == "sensitivity" || c == "min_conf") cols.push_back({c, P_FLOAT, true});
    }
    const size_t n = dets.size();
    for (const auto &d : dets) {
        put_f32(cols[0].values, d.start_time);
        put_f32(cols[1].values, d.end_time);
        put_str(cols[2].values, d.scientific_name);
        put_str(cols[3].values, d.common_name);
        put_f32(cols[4].values, d.confidence);
        std::string name;
        put_str(cols[5].values, path_file_name(d.file_path, name) ? name : d.file_path);
    }
    for (size_t c = 6; c < cols.size(); c++) cols[c].nulls = n;   // DetectionMetadata is all-None on this path

    std::string file = "PAR1";
    struct Chunk { int64_t offset, comp, uncomp; };
    std::vector<Chunk> chunks;
    if (n > 0)
        for (auto &c : cols) {
            // definition levels (optional columns only): one RLE run of n zeros / ones, bit width 1, no length prefix in V2
            std::string levels;
            if (c.optional) {
                Thrift t; t.varint((uint64_t)n << 1);
                levels = t.b;
                levels.push_back((char)(c.nulls == n ? 0 : 1));
            }
            const std::string comp = snappy_literal(c.values);
            Thrift h;
            h.last = {0};
            h.i32(1, 3);                                                  // PageType::DATA_PAGE_V2
            h.i32(2, (int32_t)(levels.size() + c.values.size()));         // uncompressed_page_size (levels included)
            h.i32(3, (int32_t)(levels.size() + comp.size()));             // compressed_page_size
            h.begin_struct(8);                                            // data_page_header_v2
            h.i32(1, (int32_t)n); h.i32(2, (int32_t)c.nulls); h.i32(3, (int32_t)n);
            h.i32(4, 0);                                                  // Encoding::PLAIN
            h.i32(5, (int32_t)levels.size()); h.i32(6, 0);
            h.boolean(7, true);
            h.end_struct();
            h.b.push_back(0);
            const int64_t off = (int64_t)file.size();
            file += h.b; file += levels; file += comp;
            chunks.push_back({off, (int64_t)(h.b.size() + levels.size() + comp.size()), (int64_t)(h.b.size() + levels.size() + c.values.size())});
        }

    Thrift m;
    m.i32(1, 2);                                                          // version (WriterVersion::PARQUET_2_0)
    m.list(2, T_STRUCT, cols.size() + 1);
    m.begin_struct_elem();                                                // root
    m.str(4, "arrow_schema");
    m.i32(5, (int32_t)cols.size());
    m.end_struct();
    for (const auto &c : cols) {
        m.begin_struct_elem();
        m.i32(1, c.type);
        m.i32(3, c.optional ? 1 : 0);                                     // FieldRepetitionType
        m.str(4, c.name);
        if (c.converted >= 0) m.i32(6, c.converted);
        if (c.logical == 1) { m.begin_struct(10); m.begin_struct(1); m.end_struct(); m.end_struct(); }                 // STRING
        if (c.logical == 10) { m.begin_struct(10); m.begin_struct(10); m.i8(1, 8); m.boolean(2, false); m.end_struct(); m.end_struct(); }   // INTEGER(8, unsigned)
        m.end_struct();
    }
    m.i64(3, (int64_t)n);
    m.list(4, T_STRUCT, n > 0 ? 1 : 0);
    if (n > 0) {
        m.begin_struct_elem();                                            // RowGroup
        m.list(1, T_STRUCT, cols.size());
        int64_t total_uncomp = 0, total_comp = 0;
        for (size_t i = 0; i < cols.size(); i++) {
            const auto &c = cols[i];
            m.begin_struct_elem();                                        // ColumnChunk
            m.i64(2, chunks[i].offset);
            m.begin_struct(3);                                            // ColumnMetaData
            m.i32(1, c.type);
            m.list(2, T_I32, 2); m.zz(0); m.zz(3);                         // encodings: PLAIN, RLE (levels)
            m.list(3, T_BINARY, 1); m.varint(c.name.size()); m.b += c.name;
            m.i32(4, 1);                                                  // CompressionCodec::SNAPPY
            m.i64(5, (int64_t)n);
            m.i64(6, chunks[i].uncomp);
            m.i64(7, chunks[i].comp);
            m.i64(9, chunks[i].offset);                                   // data_page_offset
            m.end_struct();
            m.end_struct();
            total_uncomp += chunks[i].uncomp; total_comp += chunks[i].comp;
        }
        m.i64(2, total_uncomp);
        m.i64(3, (int64_t)n);
        m.i64(5, chunks[0].offset);
        m.i64(6, total_comp);
        m.end_struct();
    }
    m.str(6, "birda-hip (libbirda_hip.so host writer)");
    m.b.push_back(0);
    file += m.b;
    const uint32_t flen = (uint32_t)m.b.size();
    char lb[4]; memcpy(lb, &flen, 4);
    file.append(lb, 4);
    file += "PAR1";

    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "ParquetFileCreate: cannot create " + path; return BH_ERR_IO; }
    const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size() && fflush(f) == 0;
    if (fclose(f) != 0 || !ok) { err = "ParquetWrite: write failed: " + path; return BH_ERR_IO; }
    return BH_OK;
}

}  // namespace bhh
