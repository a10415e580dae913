// obj_loader.cpp — Wavefront OBJ / MTL reader of the headless host layer (see obj_loader.h).
//
// The reference feeds its Model from tinyobjloader (src/Model.cpp:54, triangulation on, default-colour fallback
// on); what the ray tracer sees — vertex positions, which corners form which triangle, material per triangle —
// has to be the same here, bit for bit, for the same file. This reader is organised differently from that library
// (whole file in memory, a cursor over one line, numbers scanned into their decimal pieces and then evaluated,
// statements and MTL keys looked up in tables, a polygon-clipping class), but it keeps the behaviours that decide
// the output; tests/test_host_cpu.py compares it with the library's output on the files under tests/golden/obj*:
//
//   numbers    integer digits accumulate as m = m*10 + d in double; the k-th fraction digit adds d * 10^-k (a table up
//              to 10^-7, pow beyond); an exponent e scales by ldexp(m * 5^e, e); the double is then narrowed to
//              float. ".5", "-.5", "1e", "1e+" are not numbers (the caller's default is used); characters after a
//              valid prefix are ignored ("1.5x" = 1.5).
//   v          x y z [r g b]; without a full colour triple the vertex is white.
//   f          corners i, i/t, i//n, i/t/n; 0 is an error, negative counts back from the elements read so far. Fewer
//              than 3 corners: dropped. More than 3: ears are cut off in the plane chosen from the first
//              non-degenerate corner, always trying the same corner again after a cut (that fixes which triangles
//              come out).
//   usemtl     a *different* material id flushes the pending faces into the current shape. The name is the rest of
//              the line, blanks included. Unknown name = -1.
//   g / o      close the current shape. `g` keeps it when it has triangles, `o` when anything was pending. A `g`
//              name is the words after it joined by one blank; an `o` name is the rest of the line.
//   l / p      parsed for errors only; they keep a shape "pending" like faces do.
//   mtllib     names split at single blanks; the first file that opens is read.
//   MTL        lines are right-trimmed (OBJ lines are not). `d` wins over `Tr` (= 1 - value). The material being
//              read is always stored at the end of the file, even if it never got a name; at `newmtl` only if it
//              has one. Texture statements: options with their arguments are skipped, the rest of the line is the
//              file name.
#include "obj_loader.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iterator>
#include <limits>
#include <map>
#include <sstream>
#include <string_view>

namespace prt_host {

namespace {

using sv = std::string_view;

constexpr bool blank(char c) { return c == ' ' || c == '\t'; }
constexpr bool digit(char c) { return c >= '0' && c <= '9'; }

// ---------------------------------------------------------------------------------------------------------------
// text: the whole file, handed out line by line ("\n", "\r\n" and a lone "\r" all end a line; a NUL ends the
// useful part of a line)
// ---------------------------------------------------------------------------------------------------------------
class Lines
{
  public:
    explicit Lines(std::istream& is)
    {
        try
        {
            buf_.assign(std::istreambuf_iterator<char>(is), std::istreambuf_iterator<char>());
        }
        catch(const std::ios_base::failure&)
        {
            // e.g. a directory opened as a file: reads fail; what was read so far is the file
        }
    }
    bool next(sv& line)
    {
        if(at_ >= buf_.size()) return false;
        size_t e = at_;
        while(e < buf_.size() && buf_[e] != '\n' && buf_[e] != '\r') ++e;
        line = sv(buf_.data() + at_, e - at_);
        if(e < buf_.size()) e += (buf_[e] == '\r' && e + 1 < buf_.size() && buf_[e + 1] == '\n') ? 2 : 1;
        at_ = e;
        ++number_;
        const size_t nul = line.find('\0');
        if(nul != sv::npos) line = line.substr(0, nul);
        return true;
    }
    size_t number() const { return number_; }

  private:
    std::string buf_;
    size_t      at_ = 0, number_ = 0;
};

// a read position inside one line
class Cursor
{
  public:
    explicit Cursor(sv s) : s_(s) {}
    bool done() const { return s_.empty(); }
    void skipBlanks()
    {
        size_t k = 0;
        while(k < s_.size() && blank(s_[k])) ++k;
        s_.remove_prefix(k);
    }
    void skip(size_t n) { s_.remove_prefix(n < s_.size() ? n : s_.size()); }
    sv   rest() const { return s_; }
    // the next run of non-blank characters (after any blanks); empty at the end of the line
    sv word()
    {
        skipBlanks();
        size_t k = 0;
        while(k < s_.size() && !blank(s_[k])) ++k;
        const sv w = s_.substr(0, k);
        s_.remove_prefix(k);
        return w;
    }
    sv peekWord() const
    {
        Cursor t(*this);
        return t.word();
    }
    // statement keyword: the line's first word, which must be followed by a blank (a bare keyword is no statement)
    sv keyword() const
    {
        size_t k = 0;
        while(k < s_.size() && !blank(s_[k])) ++k;
        return k < s_.size() ? s_.substr(0, k) : sv();
    }

  private:
    sv s_;
};

// ---------------------------------------------------------------------------------------------------------------
// numbers
// ---------------------------------------------------------------------------------------------------------------
struct Decimal  // the pieces of  [sign] digits [. digits] [e [sign] digits]
{
    bool negative = false, expNegative = false;
    sv   whole, fraction, exponent;
};

// Splits the longest valid prefix of `t` into its pieces. False when `t` does not start with a number.
bool scan_decimal(sv t, Decimal& d)
{
    size_t i = 0;
    auto   run = [&](sv& out) {
        const size_t b = i;
        while(i < t.size() && digit(t[i])) ++i;
        out = t.substr(b, i - b);
    };
    if(t.empty()) return false;
    if(t[0] == '+' || t[0] == '-')
    {
        d.negative = t[0] == '-';
        i          = 1;
    }
    run(d.whole);
    if(d.whole.empty()) return false;
    if(i < t.size() && t[i] == '.')
    {
        ++i;
        run(d.fraction);
    }
    if(i < t.size() && (t[i] == 'e' || t[i] == 'E'))
    {
        ++i;
        if(i < t.size() && (t[i] == '+' || t[i] == '-'))
        {
            d.expNegative = t[i] == '-';
            ++i;
        }
        run(d.exponent);
        if(d.exponent.empty()) return false;  // "1e", "1e+": not a number at all
    }
    return true;
}

double decimal_value(const Decimal& d)
{
    static const double tenth[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
    double m = 0.0;
    for(char c : d.whole) m = m * 10 + (c - '0');
    int place = 1;
    for(char c : d.fraction)
    {
        m += (c - '0') * (place < 8 ? tenth[place] : std::pow(10.0, -place));
        ++place;
    }
    int e = 0;
    for(char c : d.exponent) e = e * 10 + (c - '0');
    if(d.expNegative) e = -e;
    const double mag = e ? std::ldexp(m * std::pow(5.0, e), e) : m;
    return (d.negative ? -1 : 1) * mag;
}

// the next word as a float; `fallback` when it is not a number (the word is consumed either way)
float real_or(Cursor& c, double fallback = 0.0)
{
    Decimal d;
    double  v = fallback;
    if(scan_decimal(c.word(), d)) v = decimal_value(d);
    return (float)v;
}
bool real_if(Cursor& c, float& out)
{
    Decimal d;
    if(!scan_decimal(c.word(), d)) return false;
    out = (float)decimal_value(d);
    return true;
}
// leading integer of `t` ("12abc" = 12, "abc" = 0, "" = 0), as the C library's atoi() converts it: the digits
// saturate at the range of long and the result is cut to the low 32 bits
int leading_int(sv t)
{
    size_t i   = 0;
    bool   neg = false;
    if(i < t.size() && (t[i] == '+' || t[i] == '-')) neg = t[i++] == '-';
    const unsigned long long top = neg ? 0x8000000000000000ull : 0x7fffffffffffffffull;
    unsigned long long       v   = 0;
    for(; i < t.size() && digit(t[i]); ++i)
    {
        const unsigned d = (unsigned)(t[i] - '0');
        v                = v > (top - d) / 10 ? top : v * 10 + d;
    }
    return (int)(unsigned)(neg ? 0ull - v : v);
}

// ---------------------------------------------------------------------------------------------------------------
// face corners
// ---------------------------------------------------------------------------------------------------------------
struct Counts
{
    int v, vn, vt;
};
// 1-based or negative (relative) file index -> 0-based; 0 is not an index
bool resolve(int fileIndex, int count, int& out)
{
    if(fileIndex == 0) return false;
    out = fileIndex > 0 ? fileIndex - 1 : count + fileIndex;
    return true;
}
// "i", "i/t", "i//n", "i/t/n". A corner that ends in '/' takes the missing number from the word that follows
// (`after`), which is still read as a corner of its own afterwards: the loader the reference uses converts with
// atoi(), and atoi() skips the blanks.
bool parse_corner(sv w, sv after, const Counts& n, ObjIndex& out)
{
    sv  field[3];
    int nf = 0;
    for(;;)
    {
        const size_t slash = w.find('/');
        if(nf == 3) return false;  // a fourth field
        field[nf++] = w.substr(0, slash);
        if(slash == sv::npos) break;
        w.remove_prefix(slash + 1);
    }
    if(nf > 1 && field[nf - 1].empty()) field[nf - 1] = after;
    out = ObjIndex{-1, -1, -1};
    if(!resolve(leading_int(field[0]), n.v, out.vertex_index)) return false;
    if(nf == 1) return true;
    if(nf == 2) return resolve(leading_int(field[1]), n.vt, out.texcoord_index);
    // three fields: an empty middle one is the "i//n" form
    if(!(nf == 3 && field[1].empty()) && !resolve(leading_int(field[1]), n.vt, out.texcoord_index)) return false;
    return resolve(leading_int(field[2]), n.vn, out.normal_index);
}

// ---------------------------------------------------------------------------------------------------------------
// polygons -> triangles
// ---------------------------------------------------------------------------------------------------------------
using Polygon = std::vector<ObjIndex>;

class EarCutter
{
  public:
    EarCutter(const Polygon& poly, const std::vector<float>& positions) : ring_(poly), pos_(positions)
    {
        choosePlane(poly);
        area_ = signedArea(poly);
    }

    template <class Emit>
    void run(Emit&& emit)
    {
        // `at` = first corner of the candidate ear; after a cut the same position is tried again. `patience` counts
        // the candidates tried since the last cut: one full round without a cut ends the attempt.
        size_t at = 0, patience = ring_.size(), lastSize = ring_.size();
        while(ring_.size() > 3 && patience > 0)
        {
            const size_t n = ring_.size();
            if(at >= n) at -= n;
            if(lastSize != n)
            {
                lastSize = n;
                patience = n;
            }
            else
                --patience;
            if(!isEar(at))
            {
                ++at;
                continue;
            }
            emit(ring_[at % n], ring_[(at + 1) % n], ring_[(at + 2) % n]);
            ring_.erase(ring_.begin() + (ptrdiff_t)((at + 1) % n));
        }
        if(ring_.size() == 3) emit(ring_[0], ring_[1], ring_[2]);
    }

  private:
    struct P2
    {
        float x, y;
    };
    bool has(const ObjIndex& c) const
    {
        const size_t b = (size_t)c.vertex_index * 3;
        return b + ax_[0] < pos_.size() && b + ax_[1] < pos_.size();
    }
    P2 flat(const ObjIndex& c) const
    {
        const size_t b = (size_t)c.vertex_index * 3;
        return P2{pos_[b + ax_[0]], pos_[b + ax_[1]]};
    }
    // drop the axis along which the first non-degenerate corner's normal is longest (ties and a fully degenerate
    // polygon resolve as the branches below say)
    void choosePlane(const Polygon& p)
    {
        ax_[0] = 1;
        ax_[1] = 2;
        const size_t n = p.size();
        for(size_t k = 0; k < n; ++k)
        {
            const size_t a = (size_t)p[k].vertex_index * 3, b = (size_t)p[(k + 1) % n].vertex_index * 3,
                         c = (size_t)p[(k + 2) % n].vertex_index * 3;
            if(a + 2 >= pos_.size() || b + 2 >= pos_.size() || c + 2 >= pos_.size()) continue;
            const float ux = pos_[b] - pos_[a], uy = pos_[b + 1] - pos_[a + 1], uz = pos_[b + 2] - pos_[a + 2];
            const float wx = pos_[c] - pos_[b], wy = pos_[c + 1] - pos_[b + 1], wz = pos_[c + 2] - pos_[b + 2];
            const float nx = std::fabs(uy * wz - uz * wy), ny = std::fabs(uz * wx - ux * wz),
                        nz = std::fabs(ux * wy - uy * wx);
            const float tiny = std::numeric_limits<float>::epsilon();
            if(!(nx > tiny || ny > tiny || nz > tiny)) continue;
            if(nx > ny && nx > nz) return;  // project along x: keep (y, z)
            ax_[0] = 0;
            if(nz > nx && nz > ny) ax_[1] = 1;  // along z: (x, y); otherwise along y: (x, z)
            return;
        }
    }
    float signedArea(const Polygon& p) const
    {
        float        a = 0;
        const size_t n = p.size();
        for(size_t k = 0; k < n; ++k)
        {
            const ObjIndex &c0 = p[k], &c1 = p[(k + 1) % n];
            if(!has(c0) || !has(c1)) continue;
            const P2 q0 = flat(c0), q1 = flat(c1);
            a += (q0.x * q1.y - q0.y * q1.x) * 0.5f;
        }
        return a;
    }
    // does the edge a->b cross the horizontal ray from q towards +x ?
    static bool crosses(P2 a, P2 b, P2 q)
    {
        return ((a.y > q.y) != (b.y > q.y)) && (q.x < (b.x - a.x) * (q.y - a.y) / (b.y - a.y) + a.x);
    }
    static bool inside(const P2 t[3], P2 q)
    {
        // crossing number over the edges (t0,t2), (t1,t0), (t2,t1)
        bool in = false;
        if(crosses(t[0], t[2], q)) in = !in;
        if(crosses(t[1], t[0], q)) in = !in;
        if(crosses(t[2], t[1], q)) in = !in;
        return in;
    }
    bool isEar(size_t at) const
    {
        const size_t n = ring_.size();
        P2           t[3];
        for(size_t k = 0; k < 3; ++k)
        {
            const ObjIndex& c = ring_[(at + k) % n];
            t[k]              = has(c) ? flat(c) : P2{0.0f, 0.0f};
        }
        const float ux = t[1].x - t[0].x, uy = t[1].y - t[0].y, wx = t[2].x - t[1].x, wy = t[2].y - t[1].y;
        const float turn = ux * wy - uy * wx;
        if(turn * area_ < 0.0f) return false;  // reflex corner
        for(size_t k = 3; k < n; ++k)
        {
            const ObjIndex& c = ring_[(at + k) % n];
            if(has(c) && inside(t, flat(c))) return false;
        }
        return true;
    }

    Polygon                   ring_;
    const std::vector<float>& pos_;
    size_t                    ax_[2];
    float                     area_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// MTL
// ---------------------------------------------------------------------------------------------------------------
struct TexOption
{
    const char* name;
    int         words;  // arguments that follow it
};
const TexOption kTexOptions[] = {{"-blendu", 1}, {"-blendv", 1}, {"-clamp", 1}, {"-boost", 1},   {"-bm", 1},  {"-o", 3},
                                 {"-s", 3},      {"-t", 3},      {"-type", 1},  {"-imfchan", 1}, {"-mm", 2},  {"-colorspace", 1}};

// `text` = what follows the statement's keyword. Options (only when something follows them) and their arguments
// are skipped; the first other word starts the file name, which runs to the end of the line. No name: `name` stays.
void texture_name(sv text, std::string& name)
{
    Cursor c(text);
    for(;;)
    {
        c.skipBlanks();
        if(c.done()) return;
        const sv         key = c.keyword();
        const TexOption* opt = nullptr;
        for(const TexOption& o : kTexOptions)
            if(key == o.name) opt = &o;
        if(!opt)
        {
            name.assign(c.rest());
            return;
        }
        c.word();
        for(int k = 0; k < opt->words; ++k) c.word();
    }
}

struct RgbKey
{
    const char* key;
    float (ObjMaterial::*field)[3];
};
struct RealKey
{
    const char* key;
    float ObjMaterial::*field;
};
struct TexKey
{
    const char* key;
    std::string ObjMaterial::*field;
};
const RgbKey  kRgbKeys[]  = {{"Ka", &ObjMaterial::ambient},       {"Kd", &ObjMaterial::diffuse}, {"Ks", &ObjMaterial::specular},
                             {"Kt", &ObjMaterial::transmittance}, {"Tf", &ObjMaterial::transmittance},
                             {"Ke", &ObjMaterial::emission}};
const RealKey kRealKeys[] = {{"Ni", &ObjMaterial::ior}, {"Ns", &ObjMaterial::shininess}, {"Pr", &ObjMaterial::roughness},
                             {"Pm", &ObjMaterial::metallic}};
const TexKey  kTexKeys[]  = {{"map_Ka", &ObjMaterial::ambient_texname},
                             {"map_Kd", &ObjMaterial::diffuse_texname},
                             {"map_Ks", &ObjMaterial::specular_texname},
                             {"map_Ns", &ObjMaterial::specular_highlight_texname},
                             {"map_bump", &ObjMaterial::bump_texname},
                             {"map_Bump", &ObjMaterial::bump_texname},
                             {"bump", &ObjMaterial::bump_texname},
                             {"disp", &ObjMaterial::displacement_texname},
                             {"refl", &ObjMaterial::reflection_texname},
                             {"norm", &ObjMaterial::normal_texname}};

class MaterialFile
{
  public:
    MaterialFile(std::map<std::string, int>& ids, std::vector<ObjMaterial>& list, std::string& warn)
        : ids_(ids), list_(list), warn_(warn)
    {
    }
    void read(std::istream& is)
    {
        Lines lines(is);
        sv    line;
        while(lines.next(line))
        {
            const size_t last = line.find_last_not_of(" \t");
            line              = last == sv::npos ? sv() : line.substr(0, last + 1);
            Cursor c(line);
            c.skipBlanks();
            if(c.done() || c.rest()[0] == '#') continue;
            const sv key = c.keyword();
            if(key.empty()) continue;  // a lone word
            c.skip(key.size());        // the cursor now stands on the separator
            statement(key, c, lines.number());
        }
        store();  // the last material of the file, named or not
    }

  private:
    void store()
    {
        ids_.insert(std::make_pair(cur_.name, (int)list_.size()));
        list_.push_back(cur_);
    }
    void bothGiven(size_t lineNo)
    {
        std::ostringstream ss;
        ss << "Both `d` and `Tr` parameters defined for \"" << cur_.name << "\". Use the value of `d` for dissolve (line " << lineNo
           << " in .mtl.)\n";
        warn_ += ss.str();
    }
    void statement(sv key, Cursor& c, size_t lineNo)
    {
        if(key == "newmtl")
        {
            if(!cur_.name.empty()) store();
            cur_   = ObjMaterial();
            sawD_  = sawTr_ = false;
            c.skip(1);
            cur_.name.assign(c.rest());
            return;
        }
        for(const RgbKey& k : kRgbKeys)
            if(key == k.key)
            {
                float* dst = cur_.*(k.field);
                dst[0]     = real_or(c);
                dst[1]     = real_or(c);
                dst[2]     = real_or(c);
                return;
            }
        for(const RealKey& k : kRealKeys)
            if(key == k.key)
            {
                cur_.*(k.field) = real_or(c);
                return;
            }
        for(const TexKey& k : kTexKeys)
            if(key == k.key)
            {
                c.skip(1);
                texture_name(c.rest(), cur_.*(k.field));
                return;
            }
        if(key == "map_d")
        {
            c.skip(1);
            cur_.alpha_texname.assign(c.rest());  // kept as written when the line holds options only
            texture_name(c.rest(), cur_.alpha_texname);
        }
        else if(key == "illum")
            cur_.illum = leading_int(c.word());
        else if(key == "d")
        {
            cur_.dissolve = real_or(c);
            if(sawTr_) bothGiven(lineNo);
            sawD_ = true;
        }
        else if(key == "Tr")
        {
            if(sawD_)
                bothGiven(lineNo);
            else
                cur_.dissolve = 1.0f - real_or(c);
            sawTr_ = true;
        }
        // everything else (Ps, Pc, Pcr, aniso, anisor, map_Pr, ..., unknown keys) does not reach the ray tracer
    }

    std::map<std::string, int>& ids_;
    std::vector<ObjMaterial>&   list_;
    std::string&                warn_;
    ObjMaterial                 cur_;
    bool                        sawD_ = false, sawTr_ = false;
};

// ---------------------------------------------------------------------------------------------------------------
// OBJ
// ---------------------------------------------------------------------------------------------------------------
class ObjFile
{
  public:
    ObjFile(ObjData& out, std::string& err, std::string mtlDir) : out_(out), err_(err), mtlDir_(std::move(mtlDir))
    {
        if(!mtlDir_.empty() && mtlDir_.back() != '/') mtlDir_ += '/';
    }

    bool read(std::istream& is)
    {
        Lines lines(is);
        sv    line;
        while(lines.next(line))
        {
            lineNo_ = lines.number();
            Cursor c(line);
            c.skipBlanks();
            if(c.done() || c.rest()[0] == '#') continue;
            const sv key = c.keyword();
            if(key.empty()) continue;
            c.skip(key.size());
            if(!statement(key, c)) return false;
        }
        finish();
        return true;
    }

  private:
    using Handler = bool (ObjFile::*)(Cursor&);
    bool statement(sv key, Cursor& c)
    {
        static const struct
        {
            const char* key;
            Handler     fn;
        } table[] = {{"v", &ObjFile::onVertex},     {"vn", &ObjFile::onNormal},   {"vt", &ObjFile::onTexcoord},
                     {"f", &ObjFile::onFace},       {"l", &ObjFile::onLine},      {"p", &ObjFile::onPoints},
                     {"usemtl", &ObjFile::onUseMtl}, {"mtllib", &ObjFile::onMtlLib}, {"g", &ObjFile::onGroup},
                     {"o", &ObjFile::onObject}};
        for(const auto& e : table)
            if(key == e.key) return (this->*e.fn)(c);
        return true;  // s, t and unknown statements do not change what is ray-traced
    }

    Counts counts() const
    {
        return Counts{(int)(out_.vertices.size() / 3), (int)(out_.normals.size() / 3), (int)(out_.texcoords.size() / 2)};
    }

    bool onVertex(Cursor& c)
    {
        for(int k = 0; k < 3; ++k) out_.vertices.push_back(real_or(c));
        float      rgb[3];
        const bool coloured = real_if(c, rgb[0]) && real_if(c, rgb[1]) && real_if(c, rgb[2]);
        for(int k = 0; k < 3; ++k) out_.colors.push_back(coloured ? rgb[k] : 1.0f);
        return true;
    }
    bool onNormal(Cursor& c)
    {
        for(int k = 0; k < 3; ++k) out_.normals.push_back(real_or(c));
        return true;
    }
    bool onTexcoord(Cursor& c)
    {
        for(int k = 0; k < 2; ++k) out_.texcoords.push_back(real_or(c));
        return true;
    }
    // all corners of the rest of the line; false (with the reference loader's message) on a bad index
    bool corners(Cursor& c, char kind, Polygon* keep)
    {
        const Counts n = counts();
        for(sv w = c.word(); !w.empty(); w = c.word())
        {
            ObjIndex ix;
            if(!parse_corner(w, c.peekWord(), n, ix))
            {
                std::ostringstream ss;
                ss << "Failed parse `" << kind << "' line(e.g. zero value for " << (kind == 'f' ? "face" : "vertex") << " index. line "
                   << lineNo_ << ".)\n";
                err_ += ss.str();
                return false;
            }
            if(keep) keep->push_back(ix);
        }
        return true;
    }
    bool onFace(Cursor& c)
    {
        Polygon p;
        if(!corners(c, 'f', &p)) return false;
        for(const ObjIndex& ix : p)
        {
            top_.vertex_index   = std::max(top_.vertex_index, ix.vertex_index);
            top_.normal_index   = std::max(top_.normal_index, ix.normal_index);
            top_.texcoord_index = std::max(top_.texcoord_index, ix.texcoord_index);
        }
        pendingFaces_.push_back(std::move(p));
        return true;
    }
    bool onLine(Cursor& c)
    {
        pendingOther_ = true;
        return corners(c, 'l', nullptr);
    }
    bool onPoints(Cursor& c)
    {
        pendingOther_ = true;
        return corners(c, 'p', nullptr);
    }
    bool onUseMtl(Cursor& c)
    {
        c.skip(1);
        const auto it = materialIds_.find(std::string(c.rest()));
        const int  id = it == materialIds_.end() ? -1 : it->second;
        if(id != material_)
        {
            flush();
            pendingFaces_.clear();  // lines / points stay pending
            material_ = id;
        }
        return true;
    }
    bool onMtlLib(Cursor& c)
    {
        c.skip(1);
        std::vector<std::string> names;
        {
            std::string all(c.rest());
            size_t      b = 0;
            while(b < all.size())  // split at every single blank; a trailing blank adds nothing
            {
                const size_t e = all.find(' ', b);
                names.push_back(all.substr(b, e == std::string::npos ? e : e - b));
                if(e == std::string::npos) break;
                b = e + 1;
            }
        }
        if(names.empty())
        {
            std::ostringstream ss;
            ss << "Looks like empty filename for mtllib. Use default material (line " << lineNo_ << ".)\n";
            out_.warn += ss.str();
            return true;
        }
        for(const std::string& n : names)
        {
            const std::string path = mtlDir_ + n;
            std::ifstream     f(path.c_str());
            if(!f)
            {
                out_.warn += "Material file [ " + path + " ] not found.\n";
                continue;
            }
            MaterialFile(materialIds_, out_.materials, out_.warn).read(f);
            return true;
        }
        out_.warn += "Failed to load material file(s). Use default material.\n";
        return true;
    }
    bool onGroup(Cursor& c)
    {
        flush();
        if(!shape_.indices.empty()) out_.shapes.push_back(shape_);
        startShape();
        std::string joined;
        int         words = 0;
        for(sv w = c.word(); !w.empty(); w = c.word(), ++words)
        {
            if(words) joined += ' ';
            joined.append(w);
        }
        if(words == 0)
        {
            std::ostringstream ss;
            ss << "Empty group name. line: " << lineNo_ << "\n";
            out_.warn += ss.str();
        }
        name_ = joined;
        return true;
    }
    bool onObject(Cursor& c)
    {
        if(flush()) out_.shapes.push_back(shape_);
        startShape();
        c.skip(1);
        name_.assign(c.rest());
        return true;
    }

    void startShape()
    {
        shape_ = ObjShape();
        pendingFaces_.clear();
        pendingOther_ = false;
    }
    // pending faces -> triangles of the current shape; false when nothing (no face, line or point) was pending
    bool flush()
    {
        if(pendingFaces_.empty() && !pendingOther_) return false;
        shape_.name = name_;
        auto emit   = [&](const ObjIndex& a, const ObjIndex& b, const ObjIndex& c) {
            shape_.indices.push_back(a);
            shape_.indices.push_back(b);
            shape_.indices.push_back(c);
            shape_.material_ids.push_back(material_);
        };
        for(const Polygon& p : pendingFaces_)
            if(p.size() >= 3) EarCutter(p, out_.vertices).run(emit);
        return true;
    }
    void finish()
    {
        auto bounds = [&](int top, size_t have, const char* what) {
            if(top < (int)have) return;
            std::ostringstream ss;
            ss << what << " indices out of bounds (line " << lineNo_ << ".)\n\n";
            out_.warn += ss.str();
        };
        bounds(top_.vertex_index, out_.vertices.size() / 3, "Vertex");
        bounds(top_.normal_index, out_.normals.size() / 3, "Vertex normal");
        bounds(top_.texcoord_index, out_.texcoords.size() / 2, "Vertex texcoord");
        if(flush() || !shape_.indices.empty()) out_.shapes.push_back(shape_);
    }

    ObjData&                   out_;
    std::string&               err_;
    std::string                mtlDir_;
    std::map<std::string, int> materialIds_;
    std::vector<Polygon>       pendingFaces_;
    bool                       pendingOther_ = false;
    ObjShape                   shape_;
    std::string                name_;
    int                        material_ = -1;
    ObjIndex                   top_{-1, -1, -1};  // greatest index seen per kind
    size_t                     lineNo_ = 0;
};

}  // namespace

bool load_obj(const std::string& filename, const std::string& mtl_basedir, ObjData& out, std::string& err)
{
    out = ObjData();
    std::ifstream is(filename.c_str(), std::ios::binary);
    if(!is)
    {
        err += "Cannot open file [" + filename + "]\n";
        return false;
    }
    if(ObjFile(out, err, mtl_basedir).read(is)) return true;
    // a failed load hands back no vertex data (shapes closed before the error, materials and warnings stay)
    out.vertices.clear();
    out.normals.clear();
    out.texcoords.clear();
    out.colors.clear();
    return false;
}

}  // namespace prt_host
