// circuit.cpp -- Circuit::read (acir/src/circuit/mod.rs:154-161): gzip + bincode-1.3 default config
// (fixint little-endian, u64 lengths, u32 enum tags, u8 Option tag; FieldElement = hex String,
// acir_field/src/generic_ark.rs:114-134, parsed like from_hex :263-267, i.e. reduced mod p).
#include "circuit.hpp"
#include <algorithm>
#include <map>
#include <stdexcept>
#include <zlib.h>

namespace acvm {
namespace {

struct Reader {
    const uint8_t *p, *end;
    uint32_t max_w = 0;
    void need(size_t n) {
        if ((size_t)(end - p) < n) throw std::runtime_error("unexpected end of input");
    }
    uint8_t u8() { need(1); return *p++; }
    uint32_t u32() { need(4); uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
    uint64_t u64() { need(8); uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
    size_t len(size_t min_elem) {
        uint64_t n = u64();
        if (min_elem && n > (uint64_t)(end - p) / min_elem) throw std::runtime_error("length prefix exceeds input");
        return (size_t)n;
    }
    uint32_t witness() { uint32_t w = u32(); if (w > max_w) max_w = w; return w; }
    static int hexval(uint8_t c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    FrH fr() {
        size_t n = len(1);
        need(n);
        const uint8_t *s = p;
        p += n;
        if (n >= 2 && s[0] == '0' && s[1] == 'x') { s += 2; n -= 2; }
        // from_hex (generic_ark.rs:263-267): hex::decode of any even length, then from_be_bytes_reduce
        if (n % 2) throw std::runtime_error("bad field element hex");
        std::vector<uint8_t> b(n / 2);
        for (size_t i = 0; i < n / 2; i++) {
            int h = hexval(s[2 * i]), l = hexval(s[2 * i + 1]);
            if (h < 0 || l < 0) throw std::runtime_error("bad field element hex");
            b[i] = (uint8_t)(h * 16 + l);
        }
        return frh::from_be_bytes_reduce(b.data(), b.size());
    }
    std::string str() {
        size_t n = len(1);
        need(n);
        std::string s((const char *)p, n);
        p += n;
        return s;
    }
    Expr expr() {
        Expr e;
        size_t nm = len(16);
        e.mul.resize(nm);
        for (auto &t : e.mul) { t.c = fr(); t.l = witness(); t.r = witness(); }
        size_t nl = len(12);
        e.lin.resize(nl);
        for (auto &t : e.lin) { t.c = fr(); t.w = witness(); }
        e.qc = fr();
        return e;
    }
    bool opt_expr(Expr &e) {
        uint8_t tag = u8();
        if (tag == 0) return false;
        if (tag != 1) throw std::runtime_error("bad Option tag");
        e = expr();
        return true;
    }
    FuncInput finput() { FuncInput f; f.witness = witness(); f.num_bits = u32(); return f; }
    std::vector<FuncInput> finput_vec() {
        size_t n = len(8);
        std::vector<FuncInput> v(n);
        for (auto &f : v) f = finput();
        return v;
    }
    std::vector<uint32_t> witness_vec() {
        size_t n = len(4);
        std::vector<uint32_t> v(n);
        for (auto &w : v) w = witness();
        return v;
    }
    std::unique_ptr<BlackBoxCall> blackbox() {
        auto b = std::make_unique<BlackBoxCall>();
        b->func = u32();
        switch (b->func) {
        case BB_AND: case BB_XOR:
            b->in[0] = {finput()}; b->in[1] = {finput()}; b->out = {witness()};
            break;
        case BB_RANGE: b->in[0] = {finput()}; break;
        case BB_SHA256: case BB_BLAKE2S: case BB_KECCAK256:
            b->in[0] = finput_vec(); b->out = witness_vec();
            break;
        case BB_SCHNORR_VERIFY:
            b->in[0] = {finput()}; b->in[1] = {finput()}; b->in[2] = finput_vec(); b->in[3] = finput_vec();
            b->out = {witness()};
            break;
        case BB_PEDERSEN: {
            b->in[0] = finput_vec(); b->domain_separator = u32();
            uint32_t x = witness(), y = witness();
            b->out = {x, y};
            break;
        }
        case BB_HASH_TO_FIELD_128: b->in[0] = finput_vec(); b->out = {witness()}; break;
        case BB_ECDSA_SECP256K1: case BB_ECDSA_SECP256R1:
            for (int g = 0; g < 4; g++) b->in[g] = finput_vec();
            b->out = {witness()};
            break;
        case BB_FIXED_BASE_SCALAR_MUL: {
            b->in[0] = {finput()}; b->in[1] = {finput()};
            uint32_t x = witness(), y = witness();
            b->out = {x, y};
            break;
        }
        case BB_KECCAK256_VAR:
            b->in[0] = finput_vec(); b->in[1] = {finput()}; b->out = witness_vec();
            break;
        case BB_RECURSIVE_AGGREGATION: {
            b->in[0] = finput_vec(); b->in[1] = finput_vec(); b->in[2] = finput_vec(); b->in[3] = {finput()};
            uint8_t tag = u8();
            if (tag == 1) { b->has_in_agg = true; b->in_agg = finput_vec(); }
            else if (tag != 0) throw std::runtime_error("bad Option tag");
            b->out = witness_vec();
            break;
        }
        default: throw std::runtime_error("unknown BlackBoxFuncCall tag");
        }
        return b;
    }
    std::unique_ptr<Directive> directive() {
        auto d = std::make_unique<Directive>();
        d->kind = u32();
        switch (d->kind) {
        case DIR_QUOTIENT:
            d->a = expr(); d->b = expr(); d->q = witness(); d->r = witness();
            d->has_predicate = opt_expr(d->predicate);
            break;
        case DIR_TO_LE_RADIX:
            d->a = expr(); d->bw = witness_vec(); d->radix = u32();
            break;
        case DIR_PERMUTATION_SORT: {
            size_t n = len(8);
            d->sort_inputs.resize(n);
            for (auto &t : d->sort_inputs) {
                size_t k = len(24);
                t.resize(k);
                for (auto &e : t) e = expr();
            }
            d->tuple = u32();
            d->bw = witness_vec();
            size_t m = len(4);
            d->sort_by.resize(m);
            for (auto &x : d->sort_by) x = u32();
            break;
        }
        default: throw std::runtime_error("unknown Directive tag");
        }
        return d;
    }
    RegOrMem rom() {
        RegOrMem m{u32(), 0, 0};
        m.reg = u64();
        if (m.kind == 1 || m.kind == 2) m.size = u64();
        else if (m.kind != 0) throw std::runtime_error("bad RegisterOrMemory tag");
        return m;
    }
    std::vector<RegOrMem> rom_vec() {
        size_t n = len(12);
        std::vector<RegOrMem> v(n);
        for (auto &m : v) m = rom();
        return v;
    }
    BrilligOp brillig_op() {
        BrilligOp o;
        o.op = u32();
        switch (o.op) {
        case BR_BINARY_FIELD_OP:
            o.a = u64(); o.sub_op = u32(); o.b = u64(); o.c = u64();
            if (o.sub_op > 4) throw std::runtime_error("bad BinaryFieldOp");
            break;
        case BR_BINARY_INT_OP:
            o.a = u64(); o.sub_op = u32(); o.bit_size = u32(); o.b = u64(); o.c = u64();
            if (o.sub_op > 12) throw std::runtime_error("bad BinaryIntOp");
            break;
        case BR_JUMP_IF_NOT: case BR_JUMP_IF: o.a = u64(); o.location = u64(); break;
        case BR_JUMP: case BR_CALL: o.location = u64(); break;
        case BR_CONST: o.a = u64(); o.value = fr(); break;
        case BR_RETURN: case BR_TRAP: case BR_STOP: break;
        case BR_FOREIGN_CALL: o.function = str(); o.dests = rom_vec(); o.inputs = rom_vec(); break;
        case BR_MOV: case BR_LOAD: case BR_STORE: o.a = u64(); o.b = u64(); break;
        case BR_BLACK_BOX: {
            static const int nwords[9] = {4, 4, 4, 3, 9, 9, 7, 5, 4};
            o.bbop = u32();
            if (o.bbop > 8) throw std::runtime_error("bad BlackBoxOp");
            for (int i = 0; i < nwords[o.bbop]; i++) o.bb[i] = u64();
            break;
        }
        default: throw std::runtime_error("unknown brillig opcode tag");
        }
        return o;
    }
    std::unique_ptr<BrilligCall> brillig() {
        auto b = std::make_unique<BrilligCall>();
        size_t n = len(4);
        b->inputs.resize(n);
        for (auto &in : b->inputs) {
            uint32_t tag = u32();
            if (tag == 0) in.single = expr();
            else if (tag == 1) {
                in.is_array = true;
                size_t k = len(24);
                in.arr.resize(k);
                for (auto &e : in.arr) e = expr();
            } else throw std::runtime_error("bad BrilligInputs tag");
        }
        n = len(4);
        b->outputs.resize(n);
        for (auto &o : b->outputs) {
            uint32_t tag = u32();
            if (tag == 0) o.w = witness();
            else if (tag == 1) { o.is_array = true; o.arr = witness_vec(); }
            else throw std::runtime_error("bad BrilligOutputs tag");
        }
        n = len(8);
        b->fc_results.resize(n);
        for (auto &r : b->fc_results) {
            size_t k = len(4);
            r.values.resize(k);
            for (auto &v : r.values) {
                uint32_t tag = u32();
                if (tag == 0) v.single = fr();
                else if (tag == 1) {
                    v.is_array = true;
                    size_t m = len(8);
                    v.arr.resize(m);
                    for (auto &x : v.arr) x = fr();
                } else throw std::runtime_error("bad ForeignCallOutput tag");
            }
        }
        n = len(4);
        b->bytecode.resize(n);
        for (auto &o : b->bytecode) o = brillig_op();
        b->has_predicate = opt_expr(b->predicate);
        return b;
    }
    Opcode opcode() {
        Opcode o;
        o.kind = u32();
        switch (o.kind) {
        case OP_ARITHMETIC: o.expr = expr(); break;
        case OP_BLACKBOX: o.bb = blackbox(); break;
        case OP_DIRECTIVE: o.dir = directive(); break;
        case OP_BRILLIG: o.brillig = brillig(); break;
        case OP_MEMORY_OP:
            o.block_id = u32();
            o.mem_operation = expr(); o.mem_index = expr(); o.mem_value = expr();
            o.has_predicate = opt_expr(o.predicate);
            break;
        case OP_MEMORY_INIT: o.block_id = u32(); o.init = witness_vec(); break;
        default: throw std::runtime_error("unknown Opcode tag");
        }
        return o;
    }
};

bool gunzip(const uint8_t *buf, size_t len, std::vector<uint8_t> &out) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return false;
    out.resize(std::min<size_t>(len * 8 + 1024, (size_t)256 << 20));
    // zlib counts in uInt: feed and drain in chunks of at most 1 GiB so that inputs of 4 GiB and more are not truncated
    const size_t CHUNK = (size_t)1 << 30;
    size_t fed = 0, have = 0;
    int rc;
    do {
        if (zs.avail_in == 0 && fed < len) {
            const size_t m = std::min(CHUNK, len - fed);
            zs.next_in = (Bytef *)buf + fed;
            zs.avail_in = (uInt)m;
            fed += m;
        }
        if (have == out.size()) out.resize(out.size() * 2);
        const size_t room = std::min(CHUNK, out.size() - have);
        zs.next_out = out.data() + have;
        zs.avail_out = (uInt)room;
        rc = inflate(&zs, Z_NO_FLUSH);
        have += room - zs.avail_out;
        if (rc == Z_BUF_ERROR && (zs.avail_in == 0 ? fed < len : zs.avail_out == 0)) rc = Z_OK;  // needs more input / more room
    } while (rc == Z_OK);
    inflateEnd(&zs);
    out.resize(have);
    return rc == Z_STREAM_END;
}

}  // namespace

std::unique_ptr<Circuit> circuit_from_bytes(const uint8_t *buf, size_t len, std::string &err) {
    std::vector<uint8_t> inflated;
    if (len >= 2 && buf[0] == 0x1f && buf[1] == 0x8b) {
        if (!gunzip(buf, len, inflated)) { err = "gzip stream is corrupt"; return nullptr; }
        buf = inflated.data();
        len = inflated.size();
    }
    try {
        Reader r{buf, buf + len};
        auto c = std::make_unique<Circuit>();
        c->current_witness_index = r.u32();
        size_t n = r.len(4);
        c->opcodes.reserve(n);
        for (size_t i = 0; i < n; i++) c->opcodes.push_back(r.opcode());
        c->private_parameters = r.witness_vec();
        c->public_parameters = r.witness_vec();
        c->return_values = r.witness_vec();
        size_t na = r.len(12);
        c->assert_messages.resize(na);
        for (auto &m : c->assert_messages) {
            uint32_t tag = r.u32();
            if (tag == 0) { m.is_brillig = false; m.acir_index = r.u64(); m.brillig_index = 0; }
            else if (tag == 1) { m.is_brillig = true; m.acir_index = r.u64(); m.brillig_index = r.u64(); }
            else throw std::runtime_error("bad OpcodeLocation tag");
            m.message = r.str();
        }
        if (r.p != r.end) throw std::runtime_error("trailing bytes after Circuit");
        c->max_witness = r.max_w > c->current_witness_index ? r.max_w : c->current_witness_index;
        return c;
    } catch (const std::exception &e) {
        err = std::string("malformed circuit: ") + e.what();
        return nullptr;
    }
}

bool witness_map_from_bytes(const uint8_t *buf, size_t len, std::vector<uint32_t> &ids, std::vector<uint8_t> &values_be32, std::string &err) {
    std::vector<uint8_t> inflated;
    if (len >= 2 && buf[0] == 0x1f && buf[1] == 0x8b) {
        if (!gunzip(buf, len, inflated)) { err = "gzip stream is corrupt"; return false; }
        buf = inflated.data();
        len = inflated.size();
    }
    try {
        Reader r{buf, buf + len};
        const uint64_t n = r.u64();
        if (n > len) throw std::runtime_error("witness map length exceeds the buffer");
        ids.clear();
        values_be32.clear();
        for (uint64_t i = 0; i < n; i++) {
            ids.push_back(r.u32());
            const FrH v = r.fr();
            uint64_t can[4];
            frh::to_canonical(v, can);
            for (int k = 0; k < 32; k++) values_be32.push_back((uint8_t)(can[(31 - k) / 8] >> (8 * ((31 - k) % 8))));
        }
        if (r.p != r.end) throw std::runtime_error("trailing bytes after the witness map");
    } catch (const std::exception &e) {
        err = e.what();
        return false;
    }
    return true;
}

bool witness_map_to_bytes(const uint32_t *ids, const uint8_t *values_be32, size_t n, std::vector<uint8_t> &out, std::string &err) {
    // BTreeMap order: ascending witness index; a repeated index keeps its last value (map insert)
    std::map<uint32_t, const uint8_t *> m;
    for (size_t i = 0; i < n; i++) m[ids[i]] = values_be32 + 32 * i;
    std::vector<uint8_t> raw;
    auto put64 = [&](uint64_t v) { for (int k = 0; k < 8; k++) raw.push_back((uint8_t)(v >> (8 * k))); };
    put64(m.size());
    static const char *hex = "0123456789abcdef";
    for (auto &kv : m) {
        for (int k = 0; k < 4; k++) raw.push_back((uint8_t)(kv.first >> (8 * k)));
        put64(64);
        for (int k = 0; k < 32; k++) { raw.push_back((uint8_t)hex[kv.second[k] >> 4]); raw.push_back((uint8_t)hex[kv.second[k] & 15]); }
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, Z_BEST_COMPRESSION, Z_DEFLATED, 16 + MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK) { err = "deflateInit2 failed"; return false; }
    if (raw.size() >= ((size_t)1 << 31)) { deflateEnd(&zs); err = "witness map above 2 GiB"; return false; }  // zlib counts in uInt
    out.resize(deflateBound(&zs, (uLong)raw.size()) + 32);
    zs.next_in = raw.data();
    zs.avail_in = (uInt)raw.size();
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    out.resize(out.size() - zs.avail_out);
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) { err = "deflate failed"; return false; }
    return true;
}

}  // namespace acvm
