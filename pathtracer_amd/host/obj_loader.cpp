// obj_loader.cpp — see obj_loader.h. Notable behaviour of the loader the reference uses, reproduced here:
//   * numbers are parsed by digit accumulation in double (integer part * 10 + d, fraction with a table of
//     powers of ten, exponent through ldexp(m * 5^e, e)) and then narrowed to float — not strtod;
//   * `v x y z [r g b]`: colours default to white for every vertex as soon as one vertex has none;
//   * faces: i, i/j, i//k, i/j/k, negative = relative; polygons are ear-clipped in the plane of their
//     first non-degenerate corner; a face with fewer than 3 corners is dropped;
//   * `usemtl` with a changed material flushes the faces read so far into the current shape; `g` starts a
//     new shape (kept when it has faces), `o` likewise but keeps the old shape only if faces were pending;
//   * MTL: `d` wins over `Tr` (= 1 - value), `Pr`/`Pm` are the PBR roughness / metallic, a texture statement's
//     options (-bm, -o, -s, ...) are skipped and the rest of the line is the file name; the last material of
//     a file is always emitted, even when the file has no `newmtl`.
#include "obj_loader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>

namespace prt_host {

namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t'; }
inline bool is_digit(char c) { return (unsigned)(c - '0') < 10u; }
inline bool is_newline(char c) { return c == '\r' || c == '\n' || c == '\0'; }

// reads one line ending in \n, \r\n or \r
bool get_line(std::istream& is, std::string& t)
{
    t.clear();
    if(is.peek() == EOF) return false;
    for(;;)
    {
        const int c = is.get();
        if(c == EOF) return true;
        if(c == '\n') return true;
        if(c == '\r')
        {
            if(is.peek() == '\n') is.get();
            return true;
        }
        t += (char)c;
    }
}

bool parse_double(const char* s, const char* end, double* result)
{
    if(s >= end) return false;
    double      mantissa = 0.0;
    int         exponent = 0;
    char        sign = '+', expSign = '+';
    const char* cur = s;
    int         read = 0;
    bool        more;
    if(*cur == '+' || *cur == '-')
    {
        sign = *cur;
        cur++;
    }
    else if(!is_digit(*cur))
        return false;
    more = cur != end;
    while(more && is_digit(*cur))
    {
        mantissa *= 10;
        mantissa += (int)(*cur - '0');
        cur++;
        read++;
        more = cur != end;
    }
    if(read == 0) return false;
    if(more)
    {
        bool toExp = false;
        if(*cur == '.')
        {
            cur++;
            read = 1;
            more = cur != end;
            static const double lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
            while(more && is_digit(*cur))
            {
                mantissa += (int)(*cur - '0') * (read < 8 ? lut[read] : std::pow(10.0, -read));
                read++;
                cur++;
                more = cur != end;
            }
            toExp = more;
        }
        else if(*cur == 'e' || *cur == 'E')
            toExp = true;
        if(toExp && (*cur == 'e' || *cur == 'E'))
        {
            cur++;
            more = cur != end;
            if(more && (*cur == '+' || *cur == '-'))
            {
                expSign = *cur;
                cur++;
            }
            else if(!is_digit(*cur))
                return false;
            read = 0;
            more = cur != end;
            while(more && is_digit(*cur))
            {
                exponent *= 10;
                exponent += (int)(*cur - '0');
                cur++;
                read++;
                more = cur != end;
            }
            exponent *= (expSign == '+' ? 1 : -1);
            if(read == 0) return false;
        }
    }
    *result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}

float parse_real(const char** tok, double def = 0.0)
{
    (*tok) += strspn(*tok, " \t");
    const char* end = (*tok) + strcspn(*tok, " \t\r");
    double      v   = def;
    parse_double(*tok, end, &v);
    *tok = end;
    return (float)v;
}
bool parse_real_opt(const char** tok, float* out)
{
    (*tok) += strspn(*tok, " \t");
    const char* end = (*tok) + strcspn(*tok, " \t\r");
    double      v;
    const bool  ok = parse_double(*tok, end, &v);
    if(ok) *out = (float)v;
    *tok = end;
    return ok;
}
int parse_int(const char** tok)
{
    (*tok) += strspn(*tok, " \t");
    const int i = atoi(*tok);
    (*tok) += strcspn(*tok, " \t\r");
    return i;
}
std::string parse_string(const char** tok)
{
    (*tok) += strspn(*tok, " \t");
    const size_t e = strcspn(*tok, " \t\r");
    std::string  s(*tok, (*tok) + e);
    (*tok) += e;
    return s;
}

bool fix_index(int idx, int n, int* ret)
{
    if(idx > 0) { *ret = idx - 1; return true; }
    if(idx == 0) return false;
    *ret = n + idx;
    return true;
}
bool parse_triple(const char** tok, int nv, int nvn, int nvt, ObjIndex* out)
{
    ObjIndex vi{-1, -1, -1};
    if(!fix_index(atoi(*tok), nv, &vi.vertex_index)) return false;
    (*tok) += strcspn(*tok, "/ \t\r");
    if((*tok)[0] != '/') { *out = vi; return true; }
    (*tok)++;
    if((*tok)[0] == '/')
    {
        (*tok)++;
        if(!fix_index(atoi(*tok), nvn, &vi.normal_index)) return false;
        (*tok) += strcspn(*tok, "/ \t\r");
        *out = vi;
        return true;
    }
    if(!fix_index(atoi(*tok), nvt, &vi.texcoord_index)) return false;
    (*tok) += strcspn(*tok, "/ \t\r");
    if((*tok)[0] != '/') { *out = vi; return true; }
    (*tok)++;
    if(!fix_index(atoi(*tok), nvn, &vi.normal_index)) return false;
    (*tok) += strcspn(*tok, "/ \t\r");
    *out = vi;
    return true;
}

// texture statement: skip the options, the remainder of the line is the name
void parse_texture_name(std::string* name, const char* line)
{
    const char* tok = line;
    bool        found = false;
    std::string tex;
    auto        opt = [&](const char* o) {
        const size_t n = strlen(o);
        return strncmp(tok, o, n) == 0 && is_space(tok[n]);
    };
    while(!is_newline(*tok))
    {
        tok += strspn(tok, " \t");
        if(opt("-blendu")) { tok += 8; parse_string(&tok); }
        else if(opt("-blendv")) { tok += 8; parse_string(&tok); }
        else if(opt("-clamp")) { tok += 7; parse_string(&tok); }
        else if(opt("-boost")) { tok += 7; parse_real(&tok, 1.0); }
        else if(opt("-bm")) { tok += 4; parse_real(&tok, 1.0); }
        else if(opt("-o")) { tok += 3; parse_real(&tok); parse_real(&tok); parse_real(&tok); }
        else if(opt("-s")) { tok += 3; parse_real(&tok, 1.0); parse_real(&tok, 1.0); parse_real(&tok, 1.0); }
        else if(opt("-t")) { tok += 3; parse_real(&tok); parse_real(&tok); parse_real(&tok); }
        else if(opt("-type")) { tok += 5; parse_string(&tok); }
        else if(opt("-imfchan")) { tok += 9; parse_string(&tok); }
        else if(opt("-mm")) { tok += 4; parse_real(&tok, 0.0); parse_real(&tok, 1.0); }
        else if(opt("-colorspace")) { tok += 12; parse_string(&tok); }
        else
        {
            tex = std::string(tok);
            tok += tex.length();
            found = true;
        }
    }
    if(found) *name = tex;
}

void load_mtl(std::map<std::string, int>& matMap, std::vector<ObjMaterial>& materials, std::istream& is, std::string& warn)
{
    ObjMaterial m;
    bool        hasD = false, hasTr = false;
    std::string line;
    size_t      lineNo = 0;
    while(get_line(is, line))
    {
        lineNo++;
        if(!line.empty()) line = line.substr(0, line.find_last_not_of(" \t") + 1);
        if(line.empty()) continue;
        const char* tok = line.c_str();
        tok += strspn(tok, " \t");
        if(tok[0] == '\0' || tok[0] == '#') continue;
        auto key = [&](const char* k) {
            const size_t n = strlen(k);
            return strncmp(tok, k, n) == 0 && is_space(tok[n]);
        };
        auto rgb = [&](float* dst, int skip) {
            tok += skip;
            dst[0] = parse_real(&tok);
            dst[1] = parse_real(&tok);
            dst[2] = parse_real(&tok);
        };
        if(key("newmtl"))
        {
            if(!m.name.empty())
            {
                matMap.insert(std::make_pair(m.name, (int)materials.size()));
                materials.push_back(m);
            }
            m    = ObjMaterial();
            hasD = hasTr = false;
            tok += 7;
            m.name = std::string(tok);
            continue;
        }
        if(key("Ka")) { rgb(m.ambient, 2); continue; }
        if(key("Kd")) { rgb(m.diffuse, 2); continue; }
        if(key("Ks")) { rgb(m.specular, 2); continue; }
        if(key("Kt") || key("Tf")) { rgb(m.transmittance, 2); continue; }
        if(key("Ni")) { tok += 2; m.ior = parse_real(&tok); continue; }
        if(key("Ke")) { rgb(m.emission, 2); continue; }
        if(key("Ns")) { tok += 2; m.shininess = parse_real(&tok); continue; }
        if(key("illum")) { tok += 6; m.illum = parse_int(&tok); continue; }
        if(key("d"))
        {
            tok += 1;
            m.dissolve = parse_real(&tok);
            if(hasTr)
            {
                std::ostringstream ss;
                ss << "Both `d` and `Tr` parameters defined for \"" << m.name << "\". Use the value of `d` for dissolve (line "
                   << lineNo << " in .mtl.)\n";
                warn += ss.str();
            }
            hasD = true;
            continue;
        }
        if(key("Tr"))
        {
            tok += 2;
            if(hasD)
            {
                std::ostringstream ss;
                ss << "Both `d` and `Tr` parameters defined for \"" << m.name << "\". Use the value of `d` for dissolve (line "
                   << lineNo << " in .mtl.)\n";
                warn += ss.str();
            }
            else
                m.dissolve = 1.0f - parse_real(&tok);
            hasTr = true;
            continue;
        }
        if(key("Pr")) { tok += 2; m.roughness = parse_real(&tok); continue; }
        if(key("Pm")) { tok += 2; m.metallic = parse_real(&tok); continue; }
        if(key("Ps") || key("Pc")) continue;
        if(key("Pcr") || key("aniso") || key("anisor")) continue;
        if(key("map_Ka")) { parse_texture_name(&m.ambient_texname, tok + 7); continue; }
        if(key("map_Kd")) { parse_texture_name(&m.diffuse_texname, tok + 7); continue; }
        if(key("map_Ks")) { parse_texture_name(&m.specular_texname, tok + 7); continue; }
        if(key("map_Ns")) { parse_texture_name(&m.specular_highlight_texname, tok + 7); continue; }
        if(key("map_bump") || key("map_Bump")) { parse_texture_name(&m.bump_texname, tok + 9); continue; }
        if(key("bump")) { parse_texture_name(&m.bump_texname, tok + 5); continue; }
        if(key("map_d")) { m.alpha_texname = tok + 6; parse_texture_name(&m.alpha_texname, tok + 6); continue; }
        if(key("disp")) { parse_texture_name(&m.displacement_texname, tok + 5); continue; }
        if(key("refl")) { parse_texture_name(&m.reflection_texname, tok + 5); continue; }
        if(key("norm")) { parse_texture_name(&m.normal_texname, tok + 5); continue; }
        // anything else is an unknown parameter
    }
    matMap.insert(std::make_pair(m.name, (int)materials.size()));
    materials.push_back(m);
}

struct Face
{
    std::vector<ObjIndex> idx;
};

// crossing-number point-in-polygon test (W. R. Franklin)
int pnpoly(int n, const float* vx, const float* vy, float tx, float ty)
{
    int c = 0;
    for(int i = 0, j = n - 1; i < n; j = i++)
        if(((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) c = !c;
    return c;
}

void emit_tri(ObjShape& shape, const ObjIndex& a, const ObjIndex& b, const ObjIndex& c, int material)
{
    shape.indices.push_back(a);
    shape.indices.push_back(b);
    shape.indices.push_back(c);
    shape.material_ids.push_back(material);
}

// flushes the pending faces into `shape`; false when nothing was pending
bool export_faces(ObjShape& shape, const std::vector<Face>& faces, bool otherPrims, int material, const std::string& name,
                  const std::vector<float>& v)
{
    if(faces.empty() && !otherPrims) return false;
    shape.name = name;
    for(const Face& face : faces)
    {
        size_t np = face.idx.size();
        if(np < 3) continue;
        // the two axes of the projection plane: from the first corner that is not degenerate
        size_t axes[2] = {1, 2};
        for(size_t k = 0; k < np; ++k)
        {
            const size_t i0 = (size_t)face.idx[(k + 0) % np].vertex_index, i1 = (size_t)face.idx[(k + 1) % np].vertex_index,
                         i2 = (size_t)face.idx[(k + 2) % np].vertex_index;
            if(3 * i0 + 2 >= v.size() || 3 * i1 + 2 >= v.size() || 3 * i2 + 2 >= v.size()) continue;
            const float e0x = v[i1 * 3] - v[i0 * 3], e0y = v[i1 * 3 + 1] - v[i0 * 3 + 1], e0z = v[i1 * 3 + 2] - v[i0 * 3 + 2];
            const float e1x = v[i2 * 3] - v[i1 * 3], e1y = v[i2 * 3 + 1] - v[i1 * 3 + 1], e1z = v[i2 * 3 + 2] - v[i1 * 3 + 2];
            const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z),
                        cz = std::fabs(e0x * e1y - e0y * e1x);
            const float eps = std::numeric_limits<float>::epsilon();
            if(cx > eps || cy > eps || cz > eps)
            {
                if(!(cx > cy && cx > cz))
                {
                    axes[0] = 0;
                    if(cz > cx && cz > cy) axes[1] = 1;
                }
                break;
            }
        }
        float area = 0;
        for(size_t k = 0; k < np; ++k)
        {
            const size_t i0 = (size_t)face.idx[(k + 0) % np].vertex_index, i1 = (size_t)face.idx[(k + 1) % np].vertex_index;
            if(i0 * 3 + axes[0] >= v.size() || i0 * 3 + axes[1] >= v.size() || i1 * 3 + axes[0] >= v.size()
               || i1 * 3 + axes[1] >= v.size())
                continue;
            const float v0x = v[i0 * 3 + axes[0]], v0y = v[i0 * 3 + axes[1]], v1x = v[i1 * 3 + axes[0]], v1y = v[i1 * 3 + axes[1]];
            area += (v0x * v1y - v0y * v1x) * 0.5f;
        }
        std::vector<ObjIndex> rem = face.idx;
        size_t                guess = 0, remainingIterations = face.idx.size(), prevRemaining = rem.size();
        while(rem.size() > 3 && remainingIterations > 0)
        {
            np = rem.size();
            if(guess >= np) guess -= np;
            if(prevRemaining != np)
            {
                prevRemaining       = np;
                remainingIterations = np;
            }
            else
                remainingIterations--;
            ObjIndex ind[3];
            float    vx[3], vy[3];
            for(size_t k = 0; k < 3; ++k)
            {
                ind[k]          = rem[(guess + k) % np];
                const size_t vi = (size_t)ind[k].vertex_index;
                if(vi * 3 + axes[0] >= v.size() || vi * 3 + axes[1] >= v.size())
                    vx[k] = vy[k] = 0.0f;
                else
                {
                    vx[k] = v[vi * 3 + axes[0]];
                    vy[k] = v[vi * 3 + axes[1]];
                }
            }
            const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
            const float cross = e0x * e1y - e0y * e1x;
            if(cross * area < 0.0f)
            {
                guess += 1;
                continue;
            }
            bool overlap = false;
            for(size_t other = 3; other < np; ++other)
            {
                const size_t idx = (guess + other) % np;
                if(idx >= rem.size()) continue;
                const size_t ovi = (size_t)rem[idx].vertex_index;
                if(ovi * 3 + axes[0] >= v.size() || ovi * 3 + axes[1] >= v.size()) continue;
                if(pnpoly(3, vx, vy, v[ovi * 3 + axes[0]], v[ovi * 3 + axes[1]]))
                {
                    overlap = true;
                    break;
                }
            }
            if(overlap)
            {
                guess += 1;
                continue;
            }
            emit_tri(shape, ind[0], ind[1], ind[2], material);
            size_t removed = (guess + 1) % np;
            while(removed + 1 < np)
            {
                rem[removed] = rem[removed + 1];
                removed += 1;
            }
            rem.pop_back();
        }
        if(rem.size() == 3) emit_tri(shape, rem[0], rem[1], rem[2], material);
    }
    return true;
}

}  // namespace

bool load_obj(const std::string& filename, const std::string& mtl_basedir, ObjData& out, std::string& err)
{
    out = ObjData();
    std::ifstream ifs(filename.c_str());
    if(!ifs)
    {
        err += "Cannot open file [" + filename + "]\n";
        return false;
    }
    std::string baseDir = mtl_basedir;
    if(!baseDir.empty() && baseDir[baseDir.length() - 1] != '/') baseDir += '/';

    std::vector<float>&        v = out.vertices;
    std::vector<float>&        vn = out.normals;
    std::vector<float>&        vt = out.texcoords;
    std::vector<float>&        vc = out.colors;
    std::vector<Face>          faces;
    bool                       otherPrims = false;  // `l` / `p` statements count as pending primitives
    std::string                name;
    std::map<std::string, int> matMap;
    int                        material = -1;
    ObjShape                   shape;
    bool                       allColors = true;
    int                        gv = -1, gvn = -1, gvt = -1;
    size_t                     lineNum = 0;
    std::string                line;
    while(get_line(ifs, line))
    {
        lineNum++;
        if(line.empty()) continue;
        const char* tok = line.c_str();
        tok += strspn(tok, " \t");
        if(tok[0] == '\0' || tok[0] == '#') continue;
        if(tok[0] == 'v' && is_space(tok[1]))
        {
            tok += 2;
            const float x = parse_real(&tok), y = parse_real(&tok), z = parse_real(&tok);
            float       r, g, b;
            const bool  found = parse_real_opt(&tok, &r) && parse_real_opt(&tok, &g) && parse_real_opt(&tok, &b);
            if(!found) r = g = b = 1.0f;
            allColors &= found;
            v.push_back(x); v.push_back(y); v.push_back(z);
            vc.push_back(r); vc.push_back(g); vc.push_back(b);  // default-colour fallback is on
            continue;
        }
        if(tok[0] == 'v' && tok[1] == 'n' && is_space(tok[2]))
        {
            tok += 3;
            const float x = parse_real(&tok), y = parse_real(&tok), z = parse_real(&tok);
            vn.push_back(x); vn.push_back(y); vn.push_back(z);
            continue;
        }
        if(tok[0] == 'v' && tok[1] == 't' && is_space(tok[2]))
        {
            tok += 3;
            const float x = parse_real(&tok), y = parse_real(&tok);
            vt.push_back(x); vt.push_back(y);
            continue;
        }
        if((tok[0] == 'l' || tok[0] == 'p') && is_space(tok[1]))
        {
            const char kind = tok[0];
            tok += 2;
            while(!is_newline(tok[0]))
            {
                ObjIndex vi;
                if(!parse_triple(&tok, (int)(v.size() / 3), (int)(vn.size() / 3), (int)(vt.size() / 2), &vi))
                {
                    std::ostringstream ss;
                    ss << "Failed parse `" << kind << "' line(e.g. zero value for vertex index. line " << lineNum << ".)\n";
                    err += ss.str();
                    return false;
                }
                tok += strspn(tok, " \t\r");
            }
            otherPrims = true;  // lines and points are not ray-traced; they only keep a shape alive
            continue;
        }
        if(tok[0] == 'f' && is_space(tok[1]))
        {
            tok += 2;
            tok += strspn(tok, " \t");
            Face face;
            while(!is_newline(tok[0]))
            {
                ObjIndex vi;
                if(!parse_triple(&tok, (int)(v.size() / 3), (int)(vn.size() / 3), (int)(vt.size() / 2), &vi))
                {
                    std::ostringstream ss;
                    ss << "Failed parse `f' line(e.g. zero value for face index. line " << lineNum << ".)\n";
                    err += ss.str();
                    return false;
                }
                gv  = gv > vi.vertex_index ? gv : vi.vertex_index;
                gvn = gvn > vi.normal_index ? gvn : vi.normal_index;
                gvt = gvt > vi.texcoord_index ? gvt : vi.texcoord_index;
                face.idx.push_back(vi);
                tok += strspn(tok, " \t\r");
            }
            faces.push_back(face);
            continue;
        }
        if(strncmp(tok, "usemtl", 6) == 0 && is_space(tok[6]))
        {
            tok += 7;
            const std::string nm(tok);
            int               newId = -1;
            const auto        it    = matMap.find(nm);
            if(it != matMap.end()) newId = it->second;
            if(newId != material)
            {
                export_faces(shape, faces, otherPrims, material, name, v);
                faces.clear();
                material = newId;
            }
            continue;
        }
        if(strncmp(tok, "mtllib", 6) == 0 && is_space(tok[6]))
        {
            tok += 7;
            std::vector<std::string> files;
            {
                std::stringstream ss;
                ss.str(std::string(tok));
                std::string item;
                while(std::getline(ss, item, ' ')) files.push_back(item);
            }
            if(files.empty())
            {
                std::ostringstream ss;
                ss << "Looks like empty filename for mtllib. Use default material (line " << lineNum << ".)\n";
                out.warn += ss.str();
            }
            else
            {
                bool found = false;
                for(const std::string& f : files)
                {
                    const std::string path = baseDir.empty() ? f : baseDir + f;
                    std::ifstream     ms(path.c_str());
                    if(!ms)
                    {
                        out.warn += "Material file [ " + path + " ] not found.\n";
                        continue;
                    }
                    load_mtl(matMap, out.materials, ms, out.warn);
                    found = true;
                    break;
                }
                if(!found) out.warn += "Failed to load material file(s). Use default material.\n";
            }
            continue;
        }
        if(tok[0] == 'g' && is_space(tok[1]))
        {
            export_faces(shape, faces, otherPrims, material, name, v);
            if(!shape.indices.empty()) out.shapes.push_back(shape);
            shape = ObjShape();
            faces.clear();
            otherPrims = false;
            std::vector<std::string> names;
            while(!is_newline(tok[0]))
            {
                names.push_back(parse_string(&tok));
                tok += strspn(tok, " \t\r");
            }
            if(names.size() < 2)
            {
                std::ostringstream ss;
                ss << "Empty group name. line: " << lineNum << "\n";
                out.warn += ss.str();
                name = "";
            }
            else
            {
                name = names[1];
                for(size_t i = 2; i < names.size(); ++i) name += " " + names[i];
            }
            continue;
        }
        if(tok[0] == 'o' && is_space(tok[1]))
        {
            if(export_faces(shape, faces, otherPrims, material, name, v)) out.shapes.push_back(shape);
            faces.clear();
            otherPrims = false;
            shape      = ObjShape();
            tok += 2;
            name = std::string(tok);
            continue;
        }
        // t (tags), s (smoothing groups) and unknown statements do not affect what is ray-traced
    }
    (void)allColors;  // with the fallback on, colours are kept whether or not every vertex had one
    if(gv >= (int)(v.size() / 3))
    {
        std::ostringstream ss;
        ss << "Vertex indices out of bounds (line " << lineNum << ".)\n\n";
        out.warn += ss.str();
    }
    if(gvn >= (int)(vn.size() / 3))
    {
        std::ostringstream ss;
        ss << "Vertex normal indices out of bounds (line " << lineNum << ".)\n\n";
        out.warn += ss.str();
    }
    if(gvt >= (int)(vt.size() / 2))
    {
        std::ostringstream ss;
        ss << "Vertex texcoord indices out of bounds (line " << lineNum << ".)\n\n";
        out.warn += ss.str();
    }
    const bool ret = export_faces(shape, faces, otherPrims, material, name, v);
    if(ret || !shape.indices.empty()) out.shapes.push_back(shape);
    return true;
}

}  // namespace prt_host
