// Scene file readers: Wavefront .obj/.mtl geometry + materials, .json or .xml camera/light file.
// Same entry points and scene semantics as src/scene/loader.cpp:144-359, with small built-in
// parsers instead of tinyobjloader / nlohmann-json / tinyxml2.
#include "loader.hpp"

#include <cctype>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <thread>
#include <unordered_map>

using cpt::V2;
using cpt::V3;

namespace {

bool ReadFile(const std::filesystem::path &path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0, std::ios::beg);
    out.resize((size_t)n);
    if (n > 0) f.read(out.data(), n);
    return true;
}

inline void SkipSpace(const char *&p) { while (*p == ' ' || *p == '\t' || *p == '\r') ++p; }
inline bool AtEol(const char *p) { return *p == '\0' || *p == '\n'; }
// A real number the way the reference's OBJ reader takes it (tinyobjloader's parseReal /
// tryParseDouble, thirdparty/tinyobjloader/tiny_obj_loader.h:866-1006): the token runs to the next
// blank; sign, digits, fraction and decimal exponent are accumulated in double (digit by digit,
// fraction digits scaled by a table of powers of ten up to 1e-7 and pow(10, -k) beyond, exponent
// applied as ldexp(m * 5^e, e)) and the result is narrowed to float.  Anything the grammar does
// not accept ("nan", "inf", hex floats, an empty exponent) leaves the fallback value.  No
// locale, no errno, and about ten times faster than strtof on files with millions of numbers.
inline bool IsDigit(char c) { return (unsigned)(c - '0') < 10u; }
inline bool ParseRealToken(const char *s, const char *s_end, double &result) {
    if (s >= s_end) return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', exp_sign = '+';
    const char *curr = s;
    int read = 0;
    bool leading_dot = false;
    if (*curr == '+' || *curr == '-') {
        sign = *curr++;
        if (curr != s_end && *curr == '.') leading_dot = true;
    } else if (*curr == '.') {
        leading_dot = true;
    } else if (!IsDigit(*curr)) {
        return false;
    }
    if (!leading_dot) {
        while (curr != s_end && IsDigit(*curr)) {
            mantissa *= 10;
            mantissa += (int)(*curr - '0');
            ++curr;
            ++read;
        }
        if (read == 0) return false;
    }
    if (curr != s_end && *curr == '.') {
        static const double kPow[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
        // pow(10, -k) for the digits behind the table, computed once by the same library call
        static const std::array<double, 40> kPowTail = [] {
            std::array<double, 40> t{};
            for (int k = 0; k < 40; k++) t[k] = std::pow(10.0, -k);
            return t;
        }();
        ++curr;
        read = 1;
        while (curr != s_end && IsDigit(*curr)) {
            mantissa += (int)(*curr - '0') * (read < 8 ? kPow[read] : (read < 40 ? kPowTail[read] : std::pow(10.0, -read)));
            ++read;
            ++curr;
        }
    } else if (curr != s_end && !(*curr == 'e' || *curr == 'E')) {
        curr = s_end;          // trailing characters after the integer part end the number
    }
    if (curr != s_end && (*curr == 'e' || *curr == 'E')) {
        ++curr;
        if (curr != s_end && (*curr == '+' || *curr == '-')) exp_sign = *curr++;
        else if (curr == s_end || !IsDigit(*curr)) return false;
        read = 0;
        while (curr != s_end && IsDigit(*curr)) {
            if (exponent > 2147483647 / 10) return false;
            exponent = exponent * 10 + (int)(*curr - '0');
            ++curr;
            ++read;
        }
        if (exp_sign == '-') exponent = -exponent;
        if (read == 0) return false;
    }
    result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}
inline float ParseFloat(const char *&p, float fallback = 0.0f) {
    while (*p == ' ' || *p == '\t') ++p;
    const char *end = p;
    while (*end != '\0' && *end != '\n' && *end != ' ' && *end != '\t' && *end != '\r') ++end;
    double v = fallback;
    ParseRealToken(p, end, v);
    p = end;
    return (float)v;
}
// decimal integer with optional sign; false when no digit follows
inline bool ParseInt(const char *&p, long &out) {
    const char *q = p;
    bool neg = false;
    if (*q == '+' || *q == '-') neg = *q++ == '-';
    if (!IsDigit(*q)) return false;
    long v = 0;
    while (IsDigit(*q)) v = v * 10 + (*q++ - '0');
    out = neg ? -v : v;
    p = q;
    return true;
}
inline std::string ParseRest(const char *&p) {
    SkipSpace(p);
    const char *b = p;
    while (!AtEol(p)) ++p;
    const char *e = p;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
    return std::string(b, e);
}
inline void NextLine(const char *&p) { while (*p && *p != '\n') ++p; if (*p == '\n') ++p; }
inline bool Keyword(const char *&p, const char *kw) {
    size_t n = std::strlen(kw);
    if (std::strncmp(p, kw, n) == 0 && (p[n] == ' ' || p[n] == '\t')) { p += n; return true; }
    return false;
}

struct MtlEntry {          // defaults of tinyobjloader's InitMaterial (thirdparty/tinyobjloader/tiny_obj_loader.h:1366-1368)
    std::string name;
    float diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0}, transmittance[3] = {0, 0, 0}, emission[3] = {0, 0, 0};
    float shininess = 1.0f, ior = 1.0f;
    std::string diffuse_tex, specular_tex, emissive_tex;
};

void ParseMtl(const std::filesystem::path &path, std::vector<MtlEntry> &mats, std::unordered_map<std::string, int> &index) {
    std::string text;
    if (!ReadFile(path, text)) return;     // a missing .mtl is a warning in the reference, not an error
    const char *p = text.c_str();
    MtlEntry *cur = nullptr;
    while (*p) {
        SkipSpace(p);
        if (Keyword(p, "newmtl")) {
            mats.emplace_back();
            cur = &mats.back();
            cur->name = ParseRest(p);
            index.emplace(cur->name, (int)mats.size() - 1);
        } else if (cur) {
            auto vec3 = [&](float *dst) { dst[0] = ParseFloat(p); dst[1] = ParseFloat(p); dst[2] = ParseFloat(p); };
            if (Keyword(p, "Kd")) vec3(cur->diffuse);
            else if (Keyword(p, "Ks")) vec3(cur->specular);
            else if (Keyword(p, "Ke")) vec3(cur->emission);
            else if (Keyword(p, "Kt") || Keyword(p, "Tf")) vec3(cur->transmittance);
            else if (Keyword(p, "Ni")) cur->ior = ParseFloat(p);
            else if (Keyword(p, "Ns")) cur->shininess = ParseFloat(p);
            else if (Keyword(p, "map_Kd")) cur->diffuse_tex = ParseRest(p);
            else if (Keyword(p, "map_Ks")) cur->specular_tex = ParseRest(p);
            else if (Keyword(p, "map_Ke")) cur->emissive_tex = ParseRest(p);
        }
        NextLine(p);
    }
}

struct ObjIndex {
    int v, vn, vt;
    bool operator==(const ObjIndex &o) const { return v == o.v && vn == o.vn && vt == o.vt; }
};
// (v, vn, vt) -> vertex number of one output mesh: open addressing over a power-of-two table,
// grown at 50 % load.  The reference de-duplicates with a node-based hash map
// (src/scene/loader.cpp:192-211); first-use order, which is what the mesh depends on, is the same.
class VertexTable {
public:
    // returns the vertex number of `key`, or assigns `next` to it and reports `inserted`
    uint32_t FindOrInsert(const ObjIndex &key, uint32_t next, bool &inserted) {
        if ((count_ + 1) * 2 > slots_.size()) Grow();
        size_t i = Hash(key) & (slots_.size() - 1);
        for (;;) {
            Slot &s = slots_[i];
            if (s.value == kEmpty) { s.key = key; s.value = next; ++count_; inserted = true; return next; }
            if (s.key == key) { inserted = false; return s.value; }
            i = (i + 1) & (slots_.size() - 1);
        }
    }
private:
    static constexpr uint32_t kEmpty = 0xffffffffu;
    struct Slot { ObjIndex key; uint32_t value = kEmpty; };
    static size_t Hash(const ObjIndex &k) {
        uint64_t h = (uint64_t)(uint32_t)k.v * 0x9e3779b97f4a7c15ull;
        h ^= ((uint64_t)(uint32_t)k.vn + 0x7f4a7c15u) * 0xc2b2ae3d27d4eb4full;
        h ^= ((uint64_t)(uint32_t)k.vt + 0x165667b1u) * 0xff51afd7ed558ccdull;
        return (size_t)(h ^ (h >> 29));
    }
public:
    void Reserve(size_t keys) {                  // room for `keys` entries without growing
        size_t want = 1024;
        while (want < keys * 2) want *= 2;
        if (want > slots_.size()) Grow(want);
    }
private:
    void Grow(size_t to = 0) {
        std::vector<Slot> old = std::move(slots_);
        slots_.assign(to ? to : (old.empty() ? 1024 : old.size() * 2), Slot{});
        count_ = 0;
        bool inserted;
        for (const Slot &s : old)
            if (s.value != kEmpty) FindOrInsert(s.key, s.value, inserted);
    }
    std::vector<Slot> slots_;
    size_t count_ = 0;
};

// one "v/vt/vn" corner as written: 1-based, or negative = counted back from the attributes read so far
bool ParseRawCorner(const char *&p, long &v, long &vt, long &vn, bool &has_vt, bool &has_vn) {
    SkipSpace(p);
    if (AtEol(p)) return false;
    has_vt = has_vn = false;
    vt = vn = 0;
    if (!ParseInt(p, v)) return false;
    if (*p == '/') {
        ++p;
        if (*p != '/' && ParseInt(p, vt)) has_vt = true;
        if (*p == '/') {
            ++p;
            if (ParseInt(p, vn)) has_vn = true;
        }
    }
    return true;
}

struct Shape {
    std::string name;
    std::vector<ObjIndex> indices;      // three per triangle
    std::vector<int> material_ids;      // one per triangle
};

// --------------------------------------------------------------------------- tiny JSON reader
struct Json {
    enum Kind { kNull, kBool, kNumber, kString, kArray, kObject } kind = kNull;
    double number = 0;
    bool boolean = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json *Get(const std::string &key) const {
        for (auto &kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JsonParser {
    const char *p;
    bool ok = true;
    void Ws() { while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') ++p; }
    bool Lit(const char *s) { size_t n = std::strlen(s); if (std::strncmp(p, s, n) == 0) { p += n; return true; } return false; }
    std::string Str() {
        std::string s;
        ++p;
        while (*p && *p != '"') {
            if (*p == '\\' && p[1]) {
                ++p;
                switch (*p) {
                    case 'n': s += '\n'; break;
                    case 't': s += '\t'; break;
                    case 'r': s += '\r'; break;
                    case 'b': s += '\b'; break;
                    case 'f': s += '\f'; break;
                    case 'u': { for (int i = 0; i < 4 && p[1]; i++) ++p; s += '?'; break; }
                    default: s += *p;
                }
                ++p;
            } else {
                s += *p++;
            }
        }
        if (*p == '"') ++p; else ok = false;
        return s;
    }
    Json Value() {
        Json j;
        Ws();
        if (*p == '{') {
            j.kind = Json::kObject;
            ++p; Ws();
            if (*p == '}') { ++p; return j; }
            while (ok) {
                Ws();
                if (*p != '"') { ok = false; break; }
                std::string k = Str();
                Ws();
                if (*p != ':') { ok = false; break; }
                ++p;
                Json v = Value();
                j.obj.emplace_back(std::move(k), std::move(v));
                Ws();
                if (*p == ',') { ++p; continue; }
                if (*p == '}') { ++p; break; }
                ok = false;
            }
        } else if (*p == '[') {
            j.kind = Json::kArray;
            ++p; Ws();
            if (*p == ']') { ++p; return j; }
            while (ok) {
                j.arr.push_back(Value());
                Ws();
                if (*p == ',') { ++p; continue; }
                if (*p == ']') { ++p; break; }
                ok = false;
            }
        } else if (*p == '"') {
            j.kind = Json::kString;
            j.str = Str();
        } else if (Lit("true")) { j.kind = Json::kBool; j.boolean = true; }
        else if (Lit("false")) { j.kind = Json::kBool; }
        else if (Lit("null")) { }
        else {
            char *end;
            j.number = std::strtod(p, &end);
            if (end == p) ok = false;
            j.kind = Json::kNumber;
            p = end;
        }
        return j;
    }
};

bool JsonVec3(const Json *j, V3 &out) {
    if (!j || j->kind != Json::kArray || j->arr.size() < 3) return false;
    for (int i = 0; i < 3; i++) if (j->arr[i].kind != Json::kNumber) return false;
    out = V3{(float)j->arr[0].number, (float)j->arr[1].number, (float)j->arr[2].number};
    return true;
}

// --------------------------------------------------------------------------- tiny XML reader
struct XmlElement {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<XmlElement> children;
    const char *Attribute(const char *key) const {
        for (auto &a : attrs) if (a.first == key) return a.second.c_str();
        return nullptr;
    }
    const XmlElement *FirstChildElement(const char *n) const {
        for (auto &c : children) if (c.name == n) return &c;
        return nullptr;
    }
};

struct XmlParser {
    const char *p;
    bool ok = true;
    void Ws() { while (*p && std::isspace((unsigned char)*p)) ++p; }
    void SkipMisc() {            // whitespace, text, comments, declarations, doctype
        for (;;) {
            while (*p && *p != '<') ++p;
            if (std::strncmp(p, "<!--", 4) == 0) { const char *e = std::strstr(p, "-->"); p = e ? e + 3 : p + std::strlen(p); continue; }
            if (std::strncmp(p, "<?", 2) == 0) { const char *e = std::strstr(p, "?>"); p = e ? e + 2 : p + std::strlen(p); continue; }
            if (std::strncmp(p, "<!", 2) == 0) { const char *e = std::strchr(p, '>'); p = e ? e + 1 : p + std::strlen(p); continue; }
            return;
        }
    }
    std::string Name() {
        const char *b = p;
        while (*p && !std::isspace((unsigned char)*p) && *p != '>' && *p != '/' && *p != '=') ++p;
        return std::string(b, p);
    }
    // parses one element starting at '<'; returns false on malformed input
    bool Element(XmlElement &el) {
        if (*p != '<') return false;
        ++p;
        el.name = Name();
        if (el.name.empty()) return false;
        for (;;) {
            Ws();
            if (*p == '/') { if (p[1] != '>') return false; p += 2; return true; }
            if (*p == '>') { ++p; break; }
            if (!*p) return false;
            std::string key = Name();
            Ws();
            if (*p != '=') return false;
            ++p; Ws();
            char q = *p;
            if (q != '"' && q != '\'') return false;
            ++p;
            const char *b = p;
            while (*p && *p != q) ++p;
            if (!*p) return false;
            el.attrs.emplace_back(std::move(key), std::string(b, p));
            ++p;
        }
        for (;;) {
            SkipMisc();
            if (!*p) return false;
            if (p[1] == '/') {
                p += 2;
                std::string close = Name();
                Ws();
                if (*p != '>' || close != el.name) return false;
                ++p;
                return true;
            }
            XmlElement child;
            if (!Element(child)) return false;
            el.children.push_back(std::move(child));
        }
    }
};

V3 Vec3FromCommaSplittedStr(const char *str) {       // src/scene/loader.cpp:99-117
    V3 vec{0.0f, 0.0f, 0.0f};
    float *v = &vec.x;
    size_t length = std::strlen(str), vec_i = 0, l = 0;
    for (size_t r = 0; r < length; r++) {
        if (str[r] == ',') {
            v[vec_i] = (float)std::atof(str + l);
            ++vec_i;
            if (vec_i == 3) break;
            l = r + 1;
        }
    }
    if (vec_i < 3) v[vec_i] = (float)std::atof(str + l);
    return vec;
}

}  // namespace

bool LoadObjScene(Scene &scene, const std::filesystem::path &path, int bsdf_type) { return LoadObjScene(scene, path, bsdf_type, nullptr); }

bool LoadObjScene(Scene &scene, const std::filesystem::path &path, int bsdf_type, std::vector<std::filesystem::path> *material_libraries) {
    std::string text;
    if (!ReadFile(path, text)) return false;
    auto base_dir = path.parent_path();

    std::vector<MtlEntry> in_materials;
    std::unordered_map<std::string, int> material_index;
    std::vector<Shape> in_shapes;
    Shape cur;
    int cur_material = -1;
    auto flush = [&]() {
        if (!cur.indices.empty()) in_shapes.push_back(std::move(cur));
        cur = Shape();
    };

    // ---- pass 1, parallel: the text is cut at line ends into one piece per thread; each piece
    // turns its lines into numbers without looking at the others.  Corner indices stay as written
    // (relative ones count back from the attribute totals at their line, known only after the
    // pieces are put together), statements that change state are kept as events in line order.
    struct RawCorner { long v, vt, vn; bool has_vt, has_vn; };
    struct Item {
        enum Kind { kFace, kUseMtl, kMtlLib, kGroup } kind;
        uint32_t first_corner = 0, num_corners = 0;         // kFace
        uint32_t v_before = 0, vt_before = 0, vn_before = 0; // attributes of this piece read before the line
        std::string text;                                    // the other kinds
    };
    struct Piece {
        const char *begin = nullptr, *end = nullptr;
        std::vector<float> V, VN, VT;
        std::vector<RawCorner> corners;
        std::vector<Item> items;
    };
    unsigned hw = std::thread::hardware_concurrency();
    size_t num_pieces = std::max<size_t>(1, std::min<size_t>({(size_t)(hw ? hw : 1), 16, text.size() / (1u << 20) + 1}));
    std::vector<Piece> pieces(num_pieces);
    {
        const char *base = text.c_str(), *stop = base + text.size();
        const char *cut = base;
        for (size_t k = 0; k < num_pieces; k++) {
            pieces[k].begin = cut;
            const char *want = k + 1 == num_pieces ? stop : base + text.size() * (k + 1) / num_pieces;
            if (want < cut) want = cut;
            while (want < stop && *want != '\n') ++want;
            if (want < stop) ++want;
            pieces[k].end = cut = want;
        }
    }
    auto parse_piece = [](Piece &pc) {
        const char *p = pc.begin;
        while (p < pc.end) {
            SkipSpace(p);
            if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
                p += 1;
                float x = ParseFloat(p), y = ParseFloat(p), z = ParseFloat(p);
                pc.V.push_back(x); pc.V.push_back(y); pc.V.push_back(z);
            } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
                p += 2;
                float x = ParseFloat(p), y = ParseFloat(p), z = ParseFloat(p);
                pc.VN.push_back(x); pc.VN.push_back(y); pc.VN.push_back(z);
            } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
                p += 2;
                float x = ParseFloat(p), y = ParseFloat(p);
                pc.VT.push_back(x); pc.VT.push_back(y);
            } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
                p += 1;
                Item it;
                it.kind = Item::kFace;
                it.first_corner = (uint32_t)pc.corners.size();
                it.v_before = (uint32_t)(pc.V.size() / 3); it.vt_before = (uint32_t)(pc.VT.size() / 2); it.vn_before = (uint32_t)(pc.VN.size() / 3);
                RawCorner c;
                while (ParseRawCorner(p, c.v, c.vt, c.vn, c.has_vt, c.has_vn)) pc.corners.push_back(c);
                it.num_corners = (uint32_t)pc.corners.size() - it.first_corner;
                pc.items.push_back(std::move(it));
            } else if (Keyword(p, "usemtl")) {
                Item it; it.kind = Item::kUseMtl; it.text = ParseRest(p); pc.items.push_back(std::move(it));
            } else if (Keyword(p, "mtllib")) {
                Item it; it.kind = Item::kMtlLib; it.text = ParseRest(p); pc.items.push_back(std::move(it));
            } else if (Keyword(p, "o") || Keyword(p, "g")) {
                Item it; it.kind = Item::kGroup; it.text = ParseRest(p); pc.items.push_back(std::move(it));
            }
            NextLine(p);
        }
    };
    {
        std::vector<std::thread> workers;
        for (size_t k = 1; k < num_pieces; k++) workers.emplace_back(parse_piece, std::ref(pieces[k]));
        parse_piece(pieces[0]);
        for (auto &w : workers) w.join();
    }

    // ---- pass 2, in file order: attribute arrays joined, indices resolved, faces triangulated
    std::vector<float> V, VN, VT;
    {
        size_t nv = 0, nn = 0, nt = 0;
        for (auto &pc : pieces) { nv += pc.V.size(); nn += pc.VN.size(); nt += pc.VT.size(); }
        V.reserve(nv); VN.reserve(nn); VT.reserve(nt);
    }
    std::vector<ObjIndex> face;
    auto fix = [](long idx, size_t n) -> int { return idx > 0 ? (int)idx - 1 : (idx < 0 ? (int)((long)n + idx) : -1); };
    for (auto &pc : pieces) {
        const size_t v0 = V.size() / 3, vt0 = VT.size() / 2, vn0 = VN.size() / 3;     // totals before this piece
        V.insert(V.end(), pc.V.begin(), pc.V.end());
        VN.insert(VN.end(), pc.VN.begin(), pc.VN.end());
        VT.insert(VT.end(), pc.VT.begin(), pc.VT.end());
        for (auto &it : pc.items) {
            if (it.kind == Item::kUseMtl) {
                auto found = material_index.find(it.text);
                cur_material = found == material_index.end() ? -1 : found->second;
            } else if (it.kind == Item::kMtlLib) {
                ParseMtl(base_dir / it.text, in_materials, material_index);
                if (material_libraries) material_libraries->push_back(base_dir / it.text);
            } else if (it.kind == Item::kGroup) {
                flush();
                cur.name = it.text;
            } else {
                // a face may only name attributes that precede it in the file (its totals at that line)
                const size_t nv = v0 + it.v_before, nvt = vt0 + it.vt_before, nvn = vn0 + it.vn_before;
                face.clear();
                for (uint32_t k = 0; k < it.num_corners; k++) {
                    const RawCorner &c = pc.corners[it.first_corner + k];
                    face.push_back(ObjIndex{fix(c.v, nv), c.has_vn ? fix(c.vn, nvn) : -1, c.has_vt ? fix(c.vt, nvt) : -1});
                }
                auto emit = [&](const ObjIndex &a, const ObjIndex &b, const ObjIndex &c3) {
                    cur.indices.push_back(a); cur.indices.push_back(b); cur.indices.push_back(c3);
                    cur.material_ids.push_back(cur_material);
                };
                bool valid = face.size() >= 3;
                for (auto &i : face) valid = valid && i.v >= 0 && (size_t)i.v < nv;
                if (valid && face.size() == 3) {
                    emit(face[0], face[1], face[2]);
                } else if (valid && face.size() == 4) {
                    // quads split along the shorter diagonal (tiny_obj_loader.h:1429-1517)
                    const float *q0 = &V[3 * face[0].v], *q1 = &V[3 * face[1].v], *q2 = &V[3 * face[2].v], *q3 = &V[3 * face[3].v];
                    float e02x = q2[0] - q0[0], e02y = q2[1] - q0[1], e02z = q2[2] - q0[2];
                    float e13x = q3[0] - q1[0], e13y = q3[1] - q1[1], e13z = q3[2] - q1[2];
                    float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
                    float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
                    if (sqr02 < sqr13) { emit(face[0], face[1], face[2]); emit(face[0], face[2], face[3]); }
                    else { emit(face[0], face[1], face[3]); emit(face[1], face[2], face[3]); }
                } else if (valid) {
                    // larger polygons: triangle fan (convex input assumed; the reference ear-clips)
                    for (size_t k = 1; k + 1 < face.size(); k++) emit(face[0], face[k], face[k + 1]);
                }
            }
        }
        pc = Piece();        // release the piece's arrays
    }
    flush();

    bool has_normal = !VN.empty();
    bool has_uv = !VT.empty();

    struct MeshData {
        std::vector<V3> positions, normals;
        std::vector<V2> texcoords;
        std::vector<uint32_t> indices;
        VertexTable index_map;
        int material_id;
        std::shared_ptr<SceneObject> object;
    };
    std::vector<std::unique_ptr<MeshData>> mesh_data;

    // one object per (shape, material) in first-appearance order; vertices de-duplicated on
    // (v, vn, vt) in first-use order (loader.cpp:171-211)
    for (auto &in_shape : in_shapes) {
        std::unordered_map<int, size_t> objects_map;
        size_t num_faces = in_shape.indices.size() / 3;
        for (size_t f = 0; f < num_faces; f++) {
            int material_id = in_shape.material_ids[f];
            auto it = objects_map.find(material_id);
            if (it == objects_map.end()) {
                std::string mat_name = material_id < 0 ? "default" : in_materials[material_id].name;
                auto md = std::make_unique<MeshData>();
                md->material_id = material_id;
                md->object = scene.AddObject(in_shape.name + "-" + mat_name);
                md->index_map.Reserve(in_shape.indices.size() / 4);      // grows further if the mesh shares few vertices
                md->indices.reserve(in_shape.indices.size());
                mesh_data.push_back(std::move(md));
                it = objects_map.insert({material_id, mesh_data.size() - 1}).first;
            }
            MeshData &data = *mesh_data[it->second];
            for (size_t i = 0; i < 3; i++) {
                const ObjIndex &in_idx = in_shape.indices[3 * f + i];
                bool inserted;
                const uint32_t vertex = data.index_map.FindOrInsert(in_idx, (uint32_t)data.positions.size(), inserted);
                if (inserted) {
                    data.positions.push_back(V3{V[3 * in_idx.v], V[3 * in_idx.v + 1], V[3 * in_idx.v + 2]});
                    if (has_normal) {
                        bool okn = in_idx.vn >= 0 && (size_t)in_idx.vn < VN.size() / 3;
                        data.normals.push_back(okn ? V3{VN[3 * in_idx.vn], VN[3 * in_idx.vn + 1], VN[3 * in_idx.vn + 2]} : V3{0.0f, 0.0f, 0.0f});
                    }
                    if (has_uv) {
                        bool okt = in_idx.vt >= 0 && (size_t)in_idx.vt < VT.size() / 2;
                        data.texcoords.push_back(okt ? V2{VT[2 * in_idx.vt], VT[2 * in_idx.vt + 1]} : V2{0.0f, 0.0f});
                    }
                }
                data.indices.push_back(vertex);
            }
        }
    }

    for (auto &dp : mesh_data) {
        MeshData &data = *dp;
        auto mesh = std::make_shared<Mesh>();
        mesh->SetPositions(std::move(data.positions));
        mesh->SetIndices(std::move(data.indices));
        if (has_normal) mesh->SetNormals(std::move(data.normals)); else mesh->CalcNormals();
        if (has_uv) mesh->SetTexcoords(std::move(data.texcoords)); else mesh->CalcTexcoords();
        data.object->AddComponent<MeshComponent>()->SetMesh(mesh);
    }

    std::vector<std::shared_ptr<Material>> materials(in_materials.size() + 1);
    for (size_t i = 0; i < in_materials.size(); i++) {
        const MtlEntry &in_mat = in_materials[i];
        auto mat = std::make_shared<Material>();
        mat->bsdf_type = bsdf_type;
        mat->name = in_mat.name;
        mat->diffuse = V3{in_mat.diffuse[0], in_mat.diffuse[1], in_mat.diffuse[2]};
        mat->specular = V3{in_mat.specular[0], in_mat.specular[1], in_mat.specular[2]};
        mat->transmittance = V3{in_mat.transmittance[0], in_mat.transmittance[1], in_mat.transmittance[2]};
        mat->emission = V3{in_mat.emission[0], in_mat.emission[1], in_mat.emission[2]};
        mat->ior = in_mat.ior;
        mat->shininess = in_mat.shininess;
        mat->opacity = (in_mat.ior == 1.0f || cpt::is_zero(mat->transmittance)) ? 1.0f : 0.0f;   // loader.cpp:243-244
        mat->diffuse_map = in_mat.diffuse_tex;
        mat->specular_map = in_mat.specular_tex;
        mat->emission_map = in_mat.emissive_tex;
        // LoadTexture, src/scene/loader.cpp:71-97; a map that cannot be read leaves the constant in place
        auto load_map = [&](std::shared_ptr<Image> &image, const std::string &name) {
            if (name.empty()) return;
            std::string error;
            image = LoadImage(base_dir / name, error);
            if (!image) std::cout << "warning: material '" << in_mat.name << "': " << error << "; the constant is used" << std::endl;
        };
        load_map(mat->diffuse_image, in_mat.diffuse_tex);
        load_map(mat->specular_image, in_mat.specular_tex);
        load_map(mat->emission_image, in_mat.emissive_tex);
        mat->SetChanged();
        materials[i] = mat;
    }
    materials[in_materials.size()] = std::make_shared<Material>();

    for (auto &dp : mesh_data) {
        size_t mat_id = dp->material_id < 0 ? in_materials.size() : (size_t)dp->material_id;
        dp->object->AddComponent<MaterialComponent>()->SetMaterial(materials[mat_id]);
    }
    return true;
}

bool LoadJsonScene(Scene &scene, const std::filesystem::path &path) {
    std::string text;
    if (!ReadFile(path, text)) return false;
    JsonParser parser{text.c_str()};
    Json root = parser.Value();
    parser.Ws();
    if (!parser.ok || *parser.p != '\0' || root.kind != Json::kObject) return false;
    const Json *camera = root.Get("camera");
    if (!camera || camera->kind != Json::kObject) return false;
    V3 pos, look_at, up;
    const Json *fov = camera->Get("fov");
    if (!JsonVec3(camera->Get("pos"), pos) || !JsonVec3(camera->Get("look_at"), look_at) || !JsonVec3(camera->Get("up"), up) || !fov ||
        fov->kind != Json::kNumber)
        return false;

    auto camera_object = scene.AddObject("camera");
    auto camera_comp = camera_object->AddComponent<CameraComponent>();
    camera_comp->pos = pos;
    camera_comp->look_at = look_at;
    camera_comp->up = up;
    camera_comp->fov = (float)fov->number;
    if (const Json *res = camera->Get("resolution")) {
        if (res->kind != Json::kArray || res->arr.size() < 2) return false;
        camera_comp->film_width = (uint32_t)res->arr[0].number;
        camera_comp->film_height = (uint32_t)res->arr[1].number;
    }
    return true;
}

bool LoadXmlScene(Scene &scene, const std::filesystem::path &path) {
    std::string text;
    if (!ReadFile(path, text)) {
        std::cout << "xml parsing error: cannot open file" << std::endl;
        return false;
    }
    // top-level elements (the reference's documents have several roots: <camera>, <light>...)
    XmlParser parser{text.c_str()};
    std::vector<XmlElement> top;
    for (;;) {
        parser.SkipMisc();
        if (!*parser.p) break;
        XmlElement el;
        if (!parser.Element(el)) {
            std::cout << "xml parsing error: malformed element" << std::endl;
            return false;
        }
        top.push_back(std::move(el));
    }
    const XmlElement *camera_node = nullptr;
    for (auto &e : top) if (e.name == "camera") { camera_node = &e; break; }
    if (!camera_node) return false;
    auto attr = [](const XmlElement *e, const char *k) -> const char * { const char *a = e ? e->Attribute(k) : nullptr; return a ? a : "0"; };
    auto vec = [&](const XmlElement *e) { return V3{(float)std::atof(attr(e, "x")), (float)std::atof(attr(e, "y")), (float)std::atof(attr(e, "z"))}; };

    auto camera_object = scene.AddObject("camera");
    auto camera_comp = camera_object->AddComponent<CameraComponent>();
    camera_comp->pos = vec(camera_node->FirstChildElement("eye"));
    camera_comp->look_at = vec(camera_node->FirstChildElement("lookat"));
    camera_comp->up = vec(camera_node->FirstChildElement("up"));
    camera_comp->fov = (float)std::atof(attr(camera_node, "fovy"));
    camera_comp->film_width = (uint32_t)std::atoi(attr(camera_node, "width"));
    camera_comp->film_height = (uint32_t)std::atoi(attr(camera_node, "height"));

    // area lights: <light mtlname=".." radiance="r,g,b"/> overrides the emission of that material
    std::unordered_map<std::string, V3> radiance_map;
    for (auto &e : top) {
        if (e.name != "light") continue;
        const char *mtlname = e.Attribute("mtlname");
        const char *radiance = e.Attribute("radiance");
        if (mtlname && radiance) radiance_map[mtlname] = Vec3FromCommaSplittedStr(radiance);
    }
    scene.ForEach<MaterialComponent>([&](MaterialComponent &material) {
        Material *mat = material.GetMaterial();
        auto it = radiance_map.find(mat->name);
        if (it != radiance_map.end()) {
            mat->emission = it->second;
            mat->emission_map.clear();
            mat->emission_image.reset();          // the constant replaces the map, as emission_map.reset() does at src/scene/loader.cpp:353
            mat->SetChanged();
        }
    });
    return true;
}
