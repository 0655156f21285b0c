// C++ mirror of the striped carrier types (zbra-core/src/striped.rs:9-83) and of the .zbra container
// writer / reader (zbra-core/src/binary.rs:56-66, 103-130, 139-174, 184-356, 359-441) over the C ABI.
//
// The writer keeps the reference's recursive descent (Table::write_to / Column::write_to /
// FieldColumn / VariantColumn) but only COLLECTS the int arrays while it walks: Int values, Double
// bit patterns, Binary / Array / Nested lengths and Enum tags (the six call sites of
// binary.rs:293-347).  One zbra_hip_encode_frames submission then encodes them all, and the frames
// `u32 count | u32 clen | E` are spliced in where the walk met them.  The reader is the mirror image:
// zbra_hip_file_index finds every int array without decoding, zbra_hip_file_decode decodes them in
// one pipelined pass, and a second walk hangs the vectors into the tree.
//
// Byte blobs (Table::Binary.data, Column::Binary.data) are stored as they are, i.e. with
// CompressionAlgorithm::None (compression.rs:238): zstd is a different path.  A tree without byte
// blobs is written under any compression config.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "binary_io.hpp"

namespace zbra {
namespace striped {

enum class Default : uint8_t { Allow = 0, Deny = 1 };  // data.rs:62-65, wire tags binary.rs:486-503

// data.rs:38-58; wire tags binary.rs:505-573
struct Encoding {
    uint8_t family = 0;  // 0 = Encoding::Int, 1 = Encoding::Binary
    uint8_t value = 0;   // IntEncoding {Int, Date, TimeSeconds, TimeMilliseconds, TimeMicroseconds} / BinaryEncoding {Binary, Utf8}
    static Encoding int_(uint8_t v = 0) { return Encoding{0, v}; }
    static Encoding binary(uint8_t v = 0) { return Encoding{1, v}; }
    bool operator==(const Encoding &o) const { return family == o.family && value == o.value; }
};

struct Column;
struct Table;
using ColumnPtr = std::shared_ptr<Column>;
using TablePtr = std::shared_ptr<Table>;

struct FieldColumn {  // striped.rs:66-70
    std::string name;
    ColumnPtr column;
};
struct VariantColumn {  // striped.rs:72-77
    std::string name;
    uint32_t tag = 0;
    ColumnPtr column;
};

// striped::Column (striped.rs:27-64): one struct with the members of every variant; `kind` says
// which are meaningful.
struct Column {
    enum Kind : uint8_t { Unit = 0, Int = 1, Double = 2, Binary = 3, Array = 4, Struct = 5, Enum = 6, Nested = 7, Reversed = 8 };
    Kind kind = Unit;
    Default default_ = Default::Allow;
    Encoding encoding;
    size_t count = 0;                     // Unit
    std::vector<int64_t> values;          // Int
    std::vector<double> doubles;          // Double
    std::vector<size_t> lengths;          // Binary, Array, Nested
    std::vector<uint8_t> data;            // Binary
    std::vector<uint32_t> tags;           // Enum
    std::vector<FieldColumn> fields;      // Struct
    std::vector<VariantColumn> variants;  // Enum
    ColumnPtr inner;                      // Array.element, Reversed.inner
    TablePtr table;                       // Nested.table

    static ColumnPtr unit(size_t n) { auto c = std::make_shared<Column>(); c->kind = Unit; c->count = n; return c; }
    static ColumnPtr int_(std::vector<int64_t> v, uint8_t int_encoding = 0) {
        auto c = std::make_shared<Column>(); c->kind = Int; c->encoding = Encoding::int_(int_encoding); c->values = std::move(v); return c;
    }
    static ColumnPtr double_(std::vector<double> v) { auto c = std::make_shared<Column>(); c->kind = Double; c->doubles = std::move(v); return c; }
    static ColumnPtr binary(std::vector<size_t> lengths, std::vector<uint8_t> data, uint8_t binary_encoding = 0) {
        auto c = std::make_shared<Column>(); c->kind = Binary; c->encoding = Encoding::binary(binary_encoding);
        c->lengths = std::move(lengths); c->data = std::move(data); return c;
    }
    static ColumnPtr array(std::vector<size_t> lengths, ColumnPtr element) {
        auto c = std::make_shared<Column>(); c->kind = Array; c->lengths = std::move(lengths); c->inner = std::move(element); return c;
    }
    static ColumnPtr struct_(std::vector<FieldColumn> fields) { auto c = std::make_shared<Column>(); c->kind = Struct; c->fields = std::move(fields); return c; }
    static ColumnPtr enum_(std::vector<uint32_t> tags, std::vector<VariantColumn> variants) {
        auto c = std::make_shared<Column>(); c->kind = Enum; c->tags = std::move(tags); c->variants = std::move(variants); return c;
    }
    static ColumnPtr nested(std::vector<size_t> lengths, TablePtr table) {
        auto c = std::make_shared<Column>(); c->kind = Nested; c->lengths = std::move(lengths); c->table = std::move(table); return c;
    }
    static ColumnPtr reversed(ColumnPtr inner) { auto c = std::make_shared<Column>(); c->kind = Reversed; c->inner = std::move(inner); return c; }
};

// striped::Table (striped.rs:9-25)
struct Table {
    enum Kind : uint8_t { Binary = 0, Array = 1, Map = 2 };
    Kind kind = Array;
    Default default_ = Default::Allow;
    Encoding encoding;            // Binary
    std::vector<uint8_t> data;    // Binary
    ColumnPtr column;             // Array.column, Map.key_column
    ColumnPtr value_column;       // Map

    static TablePtr binary(std::vector<uint8_t> data, uint8_t binary_encoding = 0) {
        auto t = std::make_shared<Table>(); t->kind = Binary; t->encoding = Encoding::binary(binary_encoding); t->data = std::move(data); return t;
    }
    static TablePtr array(ColumnPtr column) { auto t = std::make_shared<Table>(); t->kind = Array; t->column = std::move(column); return t; }
    static TablePtr map(ColumnPtr key, ColumnPtr value) {
        auto t = std::make_shared<Table>(); t->kind = Map; t->column = std::move(key); t->value_column = std::move(value); return t;
    }
};

inline bool equal(const TablePtr &a, const TablePtr &b);
inline bool equal(const ColumnPtr &a, const ColumnPtr &b) {
    if (!a || !b) return !a && !b;
    if (a->kind != b->kind) return false;
    const bool has_default = a->kind != Column::Unit && a->kind != Column::Nested && a->kind != Column::Reversed;
    if (has_default && a->default_ != b->default_) return false;
    switch (a->kind) {
        case Column::Unit: return a->count == b->count;
        case Column::Int: return a->encoding == b->encoding && a->values == b->values;
        case Column::Double:  // bit patterns, so that NaNs compare like the file does
            return a->doubles.size() == b->doubles.size() &&
                   (a->doubles.empty() || std::memcmp(a->doubles.data(), b->doubles.data(), a->doubles.size() * 8) == 0);
        case Column::Binary: return a->encoding == b->encoding && a->lengths == b->lengths && a->data == b->data;
        case Column::Array: return a->lengths == b->lengths && equal(a->inner, b->inner);
        case Column::Struct:
            if (a->fields.size() != b->fields.size()) return false;
            for (size_t i = 0; i < a->fields.size(); i++)
                if (a->fields[i].name != b->fields[i].name || !equal(a->fields[i].column, b->fields[i].column)) return false;
            return true;
        case Column::Enum:
            if (a->tags != b->tags || a->variants.size() != b->variants.size()) return false;
            for (size_t i = 0; i < a->variants.size(); i++)
                if (a->variants[i].name != b->variants[i].name || a->variants[i].tag != b->variants[i].tag ||
                    !equal(a->variants[i].column, b->variants[i].column))
                    return false;
            return true;
        case Column::Nested: return a->lengths == b->lengths && equal(a->table, b->table);
        case Column::Reversed: return equal(a->inner, b->inner);
    }
    return false;
}
inline bool equal(const TablePtr &a, const TablePtr &b) {
    if (!a || !b) return !a && !b;
    if (a->kind != b->kind || a->default_ != b->default_) return false;
    switch (a->kind) {
        case Table::Binary: return a->encoding == b->encoding && a->data == b->data;
        case Table::Array: return equal(a->column, b->column);
        case Table::Map: return equal(a->column, b->column) && equal(a->value_column, b->value_column);
    }
    return false;
}

}  // namespace striped

namespace binary {

static const char MAGIC_NUMBER[17] = "||_ZBRA||00001||";  // binary.rs:17
static const char COMPRESSION_NONE[] = "{\"binary_data\":\"None\",\"strings\":\"None\"}";  // serde JSON of CompressionConfig{None, None}

// Does the CompressionConfig JSON (compression.rs:13-41, serde's externally tagged form) say
// CompressionAlgorithm::None for `key` ("binary_data" or "strings")?  Whitespace is ignored.
inline bool config_stores_plain(const std::string &json, const char *key) {
    std::string compact;
    for (char ch : json)
        if (ch != ' ' && ch != '\t' && ch != '\n' && ch != '\r') compact.push_back(ch);
    return compact.find(std::string("\"") + key + "\":\"None\"") != std::string::npos;
}

struct Block {  // binary.rs:63-66
    uint32_t row_count = 0;
    striped::TablePtr table;
};

// The tree walk of write_to: every byte of the image except the int-array frames goes to `out`;
// int arrays are collected with the position their frame belongs at.
class TreeWriter {
public:
    // binary_data_stored / strings_stored: the config says CompressionAlgorithm::None for that kind of blob
    TreeWriter(bool binary_data_stored, bool strings_stored) : stored_{binary_data_stored, strings_stored} {}
    std::vector<uint8_t> out;
    std::vector<size_t> cuts;
    std::vector<IntArrayRef> arrays;
    bool unsupported_blob = false;  // a byte blob under a zstd config was met

    void u8(uint8_t v) { out.push_back(v); }
    void u32(uint32_t v) { write_u32(out, v); }
    void bytes(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; out.insert(out.end(), b, b + n); }
    void string(const std::string &s) { u32((uint32_t)s.size()); bytes(s.data(), s.size()); }          // write_string :599-604
    void sized(const void *p, size_t n) { u32((uint32_t)n); u32((uint32_t)n); bytes(p, n); }             // write_sized_byte_array :621-626
    void zblob(const std::vector<uint8_t> &d, int which) {                                               // write_sized_byte_array_compressed :677-687
        if (!stored_[which]) unsupported_blob = true;
        sized(d.data(), d.size());
    }
    void encoding(const striped::Encoding &e) { u8(e.family); u8(e.value); }                             // :507-519, 531-541, 556-563
    void int_array(const void *values, size_t n, bool is_u32) {
        cuts.push_back(out.size());
        arrays.push_back(IntArrayRef{values, n, is_u32});
    }

    void column(const striped::Column &c) {  // Column::write_to :275-356
        using striped::Column;
        u8((uint8_t)c.kind);
        switch (c.kind) {
            case Column::Unit: u32((uint32_t)c.count); break;
            case Column::Int: u8((uint8_t)c.default_); encoding(c.encoding); int_array(c.values.data(), c.values.size(), false); break;
            case Column::Double:  // f64::to_bits as i64: the same bytes
                u8((uint8_t)c.default_); int_array(c.doubles.data(), c.doubles.size(), false); break;
            case Column::Binary:
                u8((uint8_t)c.default_); encoding(c.encoding); int_array(c.lengths.data(), c.lengths.size(), false);  // `usize as i64`
                zblob(c.data, 1);
                break;
            case Column::Array: u8((uint8_t)c.default_); int_array(c.lengths.data(), c.lengths.size(), false); column(*c.inner); break;
            case Column::Struct:
                u8((uint8_t)c.default_); u32((uint32_t)c.fields.size());
                for (const auto &f : c.fields) { string(f.name); column(*f.column); }              // FieldColumn::write_to :446-454
                break;
            case Column::Enum:
                u8((uint8_t)c.default_); int_array(c.tags.data(), c.tags.size(), true); u32((uint32_t)c.variants.size());
                for (const auto &v : c.variants) { string(v.name); u32(v.tag); column(*v.column); }  // VariantColumn::write_to :466-475
                break;
            case Column::Nested: int_array(c.lengths.data(), c.lengths.size(), false); table(*c.table); break;
            case Column::Reversed: column(*c.inner); break;
        }
    }
    void table(const striped::Table &t) {  // Table::write_to :204-237
        using striped::Table;
        u8((uint8_t)t.kind);
        u8((uint8_t)t.default_);
        switch (t.kind) {
            case Table::Binary: encoding(t.encoding); zblob(t.data, 0); break;
            case Table::Array: column(*t.column); break;
            case Table::Map: column(*t.column); column(*t.value_column); break;
        }
    }

private:
    bool stored_[2];
};
static_assert(sizeof(size_t) == 8 && sizeof(double) == 8, "lengths and f64 bit patterns go through the i64 path as they are");

// The second walk of the reader: the tag-driven descent of read_from over an image whose int
// arrays are already decoded, in file order.
class TreeReader {
public:
    // binary_data_stored / strings_stored: the image's config says CompressionAlgorithm::None for that kind of blob
    TreeReader(const uint8_t *image, size_t len, std::vector<DecodedArray> &decoded, bool binary_data_stored = true, bool strings_stored = true)
        : r_(image, len), decoded_(decoded), stored_{binary_data_stored, strings_stored} {}
    size_t &pos() { return r_.pos; }
    bool failed = false;
    bool unsupported_blob = false;  // a byte blob under a zstd config was met: its bytes are still compressed

    uint8_t u8() { uint8_t v = 0; if (!r_.read_exact(&v, 1)) failed = true; return v; }
    uint32_t u32() { uint32_t v = 0; if (!read_u32(r_, &v)) failed = true; return v; }
    std::string string() {
        const uint32_t n = u32();
        std::string s(r_.len - r_.pos >= n ? n : 0, '\0');
        if (s.size() != n || !r_.read_exact(&s[0], n)) failed = true;
        return s;
    }
    // read_sized_byte_array_compressed :690-699.  Under CompressionAlgorithm::None the stored bytes
    // are the blob; under Zstd they would have to go through decompress_binary (compression.rs:251-268),
    // the byte codec this library does not carry: such a blob is reported, never handed back as data.
    std::vector<uint8_t> zblob(int which) {
        u32();
        const uint32_t n = u32();
        std::vector<uint8_t> d(r_.len - r_.pos >= n ? n : 0);
        if (d.size() != n || !r_.read_exact(d.data(), n)) failed = true;
        if (!stored_[which]) unsupported_blob = true;  // even an empty blob is a zstd frame there, not zero bytes
        return d;
    }
    striped::Encoding encoding() { striped::Encoding e; e.family = u8(); e.value = u8(); return e; }
    DecodedArray *next_array() {  // skips the frame in the image, hands out its decoded values
        const uint32_t count = u32(), clen = u32();
        (void)count;
        if (r_.len - r_.pos < clen || next_ >= decoded_.size()) { failed = true; return nullptr; }
        r_.pos += clen;
        return &decoded_[next_++];
    }
    template <typename T>
    std::vector<T> i64_array_as() {  // Vec<i64> reinterpreted: usize lengths, f64 bit patterns
        std::vector<T> out;
        if (DecodedArray *a = next_array()) {
            out.resize(a->i64.size());
            if (!out.empty()) std::memcpy(out.data(), a->i64.data(), out.size() * 8);
        }
        return out;
    }

    striped::ColumnPtr column() {  // Column::read_from :359-441
        using striped::Column;
        auto c = std::make_shared<Column>();
        const uint8_t tag = u8();
        if (tag > 8) { failed = true; return c; }  // BinaryError::InvalidColumnTag (the index walk has rejected it already)
        c->kind = (Column::Kind)tag;
        switch (c->kind) {
            case Column::Unit: c->count = u32(); break;
            case Column::Int:
                c->default_ = (striped::Default)u8(); c->encoding = encoding();
                if (DecodedArray *a = next_array()) c->values = std::move(a->i64);
                break;
            case Column::Double: c->default_ = (striped::Default)u8(); c->doubles = i64_array_as<double>(); break;
            case Column::Binary:
                c->default_ = (striped::Default)u8(); c->encoding = encoding(); c->lengths = i64_array_as<size_t>(); c->data = zblob(1);
                break;
            case Column::Array: c->default_ = (striped::Default)u8(); c->lengths = i64_array_as<size_t>(); c->inner = column(); break;
            case Column::Struct: {
                c->default_ = (striped::Default)u8();
                const uint32_t n = u32();
                for (uint32_t i = 0; i < n && !failed; i++) { striped::FieldColumn f; f.name = string(); f.column = column(); c->fields.push_back(std::move(f)); }
                break;
            }
            case Column::Enum: {
                c->default_ = (striped::Default)u8();
                if (DecodedArray *a = next_array()) c->tags = std::move(a->u32);
                const uint32_t n = u32();
                for (uint32_t i = 0; i < n && !failed; i++) {
                    striped::VariantColumn v; v.name = string(); v.tag = u32(); v.column = column(); c->variants.push_back(std::move(v));
                }
                break;
            }
            case Column::Nested: c->lengths = i64_array_as<size_t>(); c->table = table(); break;
            case Column::Reversed: c->inner = column(); break;
        }
        return c;
    }
    striped::TablePtr table() {  // Table::read_from :240-270
        using striped::Table;
        auto t = std::make_shared<Table>();
        const uint8_t tag = u8();
        if (tag > 2) { failed = true; return t; }
        t->kind = (Table::Kind)tag;
        t->default_ = (striped::Default)u8();
        switch (t->kind) {
            case Table::Binary: t->encoding = encoding(); t->data = zblob(0); break;
            case Table::Array: t->column = column(); break;
            case Table::Map: t->column = column(); t->value_column = column(); break;
        }
        return t;
    }

private:
    Reader r_;
    std::vector<DecodedArray> &decoded_;
    size_t next_ = 0;
    bool stored_[2];
};

// binary::BinaryFile (binary.rs:56-59) with the header's two JSON documents kept as text: the schema
// is not needed to write or read the blocks (they are self-describing by tags).
struct BinaryFile {
    std::string schema_json;
    std::string compression_json = COMPRESSION_NONE;
    std::vector<Block> blocks;

    // BinaryFile::to_bytes / write_to                                     binary.rs:96-130
    Result<std::vector<uint8_t>> to_bytes() const {
        TreeWriter w(config_stores_plain(compression_json, "binary_data"), config_stores_plain(compression_json, "strings"));
        w.bytes(MAGIC_NUMBER, 16);
        w.sized(schema_json.data(), schema_json.size());
        w.sized(compression_json.data(), compression_json.size());
        w.u32((uint32_t)blocks.size());
        for (const Block &b : blocks) {  // Block::write_to :184-192
            w.u32(b.row_count);
            w.table(*b.table);
        }
        if (w.unsupported_blob)
            return BinaryError{BinaryError::CompressionError, "byte blobs are written with CompressionAlgorithm::None only", ZBRA_ERR_INVALID_ARG};
        auto staged = encode_frames_staged(w.arrays);
        if (staged.is_err()) return staged.error();
        const StagedFrames &fr = staged.value();
        std::vector<uint8_t> image;
        image.reserve(w.out.size() + fr.offsets.back());
        size_t prev = 0;
        for (size_t i = 0; i < w.arrays.size(); i++) {
            image.insert(image.end(), w.out.begin() + prev, w.out.begin() + w.cuts[i]);
            image.insert(image.end(), fr.buf.get() + fr.offsets[i], fr.buf.get() + fr.offsets[i + 1]);
            prev = w.cuts[i];
        }
        image.insert(image.end(), w.out.begin() + prev, w.out.end());
        return image;
    }

    // BinaryFile::from_bytes / read_from                                  binary.rs:133-174
    static Result<BinaryFile> from_bytes(const uint8_t *image, size_t len) {
        auto decoded = read_file_int_arrays(image, len);
        if (decoded.is_err()) return decoded.error();
        DecodedFile df = std::move(decoded).unwrap();
        BinaryFile f;
        f.schema_json.assign((const char *)image + df.info.schema_offset, df.info.schema_len);
        f.compression_json.assign((const char *)image + df.info.compression_offset, df.info.compression_len);
        TreeReader r(image, len, df.arrays, config_stores_plain(f.compression_json, "binary_data"), config_stores_plain(f.compression_json, "strings"));
        r.pos() = df.info.blocks_offset;
        const uint32_t n = r.u32();
        for (uint32_t i = 0; i < n && !r.failed; i++) {  // Block::read_from :195-199
            Block b;
            b.row_count = r.u32();
            b.table = r.table();
            f.blocks.push_back(std::move(b));
        }
        if (r.failed) return BinaryError{BinaryError::DecompressionError, "image changed between the index walk and the tree walk", ZBRA_ERR_FORMAT};
        if (r.unsupported_blob)  // binary.rs:690-699 would call decompress_binary here
            return BinaryError{BinaryError::DecompressionError, "byte blobs compressed with zstd are not supported (CompressionAlgorithm::None only)", ZBRA_ERR_INVALID_ARG};
        return f;
    }
    static Result<BinaryFile> from_bytes(const std::vector<uint8_t> &image) { return from_bytes(image.data(), image.size()); }
};

}  // namespace binary
}  // namespace zbra
