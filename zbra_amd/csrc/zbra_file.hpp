// Host-side walker over a .zbra image: restates the control flow of BinaryFile::read_from,
// Block::read_from, Table::read_from and Column::read_from (zbra-core/src/binary.rs:139-174,
// 195-199, 240-270, 359-441) but only records where the int arrays are.  The block payload is
// self-describing through its tags, so the JSON schema is located, not parsed.
#pragma once
#include <cstring>
#include <vector>

#include "zbra_hip.h"

namespace zbra {

struct Walker {
    const uint8_t *p;
    size_t len;
    size_t pos = 0;
    bool failed = false;
    size_t fail_pos = 0;
    std::vector<zbra_file_array> *arrays = nullptr;
    size_t n_arrays = 0;
    uint64_t total_values = 0;
    uint32_t block = 0;
    int depth = 0;

    bool fail() { if (!failed) { failed = true; fail_pos = pos; } return false; }
    bool need(size_t n) { return len - pos >= n ? true : fail(); }
    bool u8(uint8_t *v) { if (!need(1)) return false; *v = p[pos++]; return true; }
    bool u32(uint32_t *v) {
        if (!need(4)) return false;
        *v = (uint32_t)p[pos] | (uint32_t)p[pos + 1] << 8 | (uint32_t)p[pos + 2] << 16 | (uint32_t)p[pos + 3] << 24;
        pos += 4;
        return true;
    }
    bool skip(size_t n) { if (!need(n)) return false; pos += n; return true; }

    bool default_tag() { uint8_t t; return u8(&t) && (t <= 1 || fail()); }          // binary.rs:496-502
    bool encoding() {                                                                // binary.rs:522-528
        uint8_t t, s;
        if (!u8(&t)) return false;
        if (t == 0) return u8(&s) && (s <= 4 || fail());                             // IntEncoding  :544-553
        if (t == 1) return u8(&s) && (s <= 1 || fail());                             // BinaryEncoding :566-571
        return fail();
    }
    bool str() { uint32_t n; return u32(&n) && skip(n); }                            // read_string :606-611
    bool zblob() { uint32_t un, cn; return u32(&un) && u32(&cn) && skip(cn); }       // :690-699 (skipped by size)
    bool int_array(uint32_t kind, uint32_t role) {                                   // :648-654
        uint32_t count, clen;
        if (!u32(&count) || !u32(&clen) || !need(clen)) return false;
        if (arrays) {
            zbra_file_array a;
            a.offset = pos; a.len = clen; a.count = count; a.kind = kind; a.role = role; a.block = block; a.reserved = 0;
            arrays->push_back(a);
        }
        n_arrays++;
        total_values += count;
        pos += clen;
        return true;
    }
    bool column() {                                                                   // binary.rs:359-441
        if (++depth > 512) return fail();  // the reference recurses without bound; stop before the stack does
        uint8_t tag;
        bool ok = u8(&tag);
        if (ok) switch (tag) {
            case 0: { uint32_t c; ok = u32(&c); break; }                                                      // Unit
            case 1: ok = default_tag() && encoding() && int_array(ZBRA_ELEM_I64, ZBRA_ROLE_INT_VALUES); break; // Int
            case 2: ok = default_tag() && int_array(ZBRA_ELEM_I64, ZBRA_ROLE_DOUBLE_BITS); break;             // Double
            case 3: ok = default_tag() && encoding() && int_array(ZBRA_ELEM_I64, ZBRA_ROLE_BINARY_LENGTHS) && zblob(); break;
            case 4: ok = default_tag() && int_array(ZBRA_ELEM_I64, ZBRA_ROLE_ARRAY_LENGTHS) && column(); break;
            case 5: {                                                                                          // Struct
                uint32_t n;
                ok = default_tag() && u32(&n);
                for (uint32_t i = 0; ok && i < n; i++) ok = str() && column();
                break;
            }
            case 6: {                                                                                          // Enum
                uint32_t n, t;
                ok = default_tag() && int_array(ZBRA_ELEM_U32, ZBRA_ROLE_ENUM_TAGS) && u32(&n);
                for (uint32_t i = 0; ok && i < n; i++) ok = str() && u32(&t) && column();
                break;
            }
            case 7: ok = int_array(ZBRA_ELEM_I64, ZBRA_ROLE_NESTED_LENGTHS) && table(); break;                 // Nested
            case 8: ok = column(); break;                                                                      // Reversed
            default: pos--; ok = fail();                                                                       // InvalidColumnTag
        }
        --depth;
        return ok;
    }
    bool table() {                                                                    // binary.rs:240-270
        if (++depth > 512) return fail();
        uint8_t tag;
        bool ok = u8(&tag);
        if (ok) switch (tag) {
            case 0: ok = default_tag() && encoding() && zblob(); break;
            case 1: ok = default_tag() && column(); break;
            case 2: ok = default_tag() && column() && column(); break;
            default: pos--; ok = fail();                                                                       // InvalidTableTag
        }
        --depth;
        return ok;
    }
    bool sized(uint64_t *off, uint64_t *n) {                                          // read_sized_byte_array :629-636
        uint32_t un, cn;
        if (!u32(&un) || !u32(&cn)) return false;
        *off = pos; *n = cn;
        return skip(cn);
    }
    bool file(zbra_file_info *info) {                                                 // binary.rs:139-174
        static const char MAGIC[17] = "||_ZBRA||00001||";                             // binary.rs:17
        if (!need(16)) return false;
        if (std::memcmp(p, MAGIC, 16) != 0) return fail();                            // InvalidMagicNumber
        pos = 16;
        if (!sized(&info->schema_offset, &info->schema_len)) return false;
        if (!sized(&info->compression_offset, &info->compression_len)) return false;
        info->blocks_offset = pos;
        if (!u32(&info->block_count)) return false;
        for (block = 0; block < info->block_count; block++) {
            uint32_t rows;
            if (!u32(&rows) || !table()) return false;                                // Block::read_from :195-199
        }
        info->n_arrays = n_arrays;
        info->total_values = total_values;
        info->end_offset = pos;
        return true;
    }
};

}  // namespace zbra
