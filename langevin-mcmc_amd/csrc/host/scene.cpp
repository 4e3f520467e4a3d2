// Scene XML -> lmc::Scene.  Follows the reference's parser key by key
// (/root/reference/src/parsescene.cpp:88-625, loadserialized.cpp:114-325, parseobj.cpp,
//  camera.cpp:12-36, envlight.cpp:24-63, scene.cpp:8-46,151-169, distribution.h:8-60).
#include "scene.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "xmlmin.h"

namespace lmc {

static const float c_PI = float(3.14159265358979323846);
static const float c_TWOPI = 2.0f * c_PI;

static float Lum(const float *v) { return v[0] * 0.212671f + v[1] * 0.715160f + v[2] * 0.072169f; }

// -------------------------------------------------------------------------------- distribution.h
void BuildPiecewise1D(const float *f, int n, std::vector<float> &func, std::vector<float> &cdf, float &funcInt) {
    func.assign(f, f + n);
    cdf.assign(n + 1, 0.f);
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
    funcInt = cdf[n];
    if (funcInt == 0.f) {
        for (int i = 1; i < n + 1; ++i) cdf[i] = float(i) / float(n);
    } else {
        for (int i = 1; i < n + 1; ++i) cdf[i] /= funcInt;
    }
}

int SampleDiscrete1D(const std::vector<float> &func, const std::vector<float> &cdf, float funcInt, float u, float *pdf) {
    int count = (int)func.size();
    const float *ptr = std::upper_bound(cdf.data(), cdf.data() + count + 1, u);
    int offset = std::min(std::max(int(ptr - cdf.data() - 1), 0), count - 1);
    if (pdf) *pdf = func[offset] / (funcInt * count);
    return offset;
}

// -------------------------------------------------------------------------------- small parsers
static std::vector<std::string> SplitList(const std::string &value) {  // regex "(,| )+" with token -1
    std::vector<std::string> out;
    std::string cur;
    bool inSep = false, any = false;
    for (char c : value) {
        if (c == ',' || c == ' ') {
            if (!inSep) {
                // std::sregex_token_iterator yields a leading empty token if the string starts with a separator
                out.push_back(cur);
                cur.clear();
                inSep = true;
            }
        } else {
            inSep = false;
            cur.push_back(c);
        }
        any = true;
    }
    if (!cur.empty() || !any) out.push_back(cur);
    return out;
}

static V3 ParseVector3(const std::string &value) {
    auto l = SplitList(value);
    V3 v;
    if (l.size() == 1) {
        v.x = v.y = v.z = std::stof(l[0]);
    } else if (l.size() == 3) {
        v.x = std::stof(l[0]), v.y = std::stof(l[1]), v.z = std::stof(l[2]);
    } else
        throw std::runtime_error("ParseVector3 failed");
    return v;
}

static M4 ParseMatrix4x4(const std::string &value) {
    auto l = SplitList(value);
    if (l.size() != 16) throw std::runtime_error("ParseMatrix4x4 failed");
    M4 m;
    int k = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) m.m[i][j] = std::stof(l[k++]);
    return m;
}

static std::string Lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), ::tolower);
    return s;
}

static M4 ParseTransform(const XmlNode &node) {
    M4 tform = M4::identity();
    for (auto &cp : node.children) {
        const XmlNode &child = *cp;
        std::string name = Lower(child.name);
        auto f = [&](const char *k, float d) { return child.has(k) ? std::stof(child.attr(k)) : d; };
        if (name == "scale") {
            if (child.has("value")) {
                float s = std::stof(child.attr("value"));
                tform = Scale(V3{s, s, s}) * tform;
            } else
                tform = Scale(V3{f("x", 1.f), f("y", 1.f), f("z", 1.f)}) * tform;
        } else if (name == "translate") {
            tform = Translate(V3{f("x", 0.f), f("y", 0.f), f("z", 0.f)}) * tform;
        } else if (name == "rotate") {
            tform = Rotate(f("angle", 0.f), V3{f("x", 0.f), f("y", 0.f), f("z", 0.f)}) * tform;
        } else if (name == "lookat") {
            tform = LookAt(ParseVector3(child.attr("origin")), ParseVector3(child.attr("target")), ParseVector3(child.attr("up"))) * tform;
        } else if (name == "matrix") {
            tform = ParseMatrix4x4(child.attr("value")) * tform;
        }
    }
    return tform;
}

static AnimXform ParseAnimatedTransform(const XmlNode &node) {
    int n = 0;
    M4 m[2] = {M4::identity(), M4::identity()};
    for (auto &c : node.children)
        if (c->name == "transform") {
            m[n++] = ParseTransform(*c);
            if (n >= 2) break;
        }
    return MakeAnimXform(m[0], m[1]);
}

// -------------------------------------------------------------------------------- meshes
namespace {
struct ZReader {  // zlib stream over a memory buffer (loadserialized.cpp:21-93)
    z_stream zs;
    ZReader(const uint8_t *p, size_t n) {
        memset(&zs, 0, sizeof(zs));
        zs.next_in = (Bytef *)p;
        zs.avail_in = (uInt)n;
        if (inflateInit2(&zs, 15) != Z_OK) throw std::runtime_error("Could not initialize ZLIB");
    }
    ~ZReader() { inflateEnd(&zs); }
    void read(void *dst, size_t size) {
        zs.next_out = (Bytef *)dst;
        zs.avail_out = (uInt)size;
        while (zs.avail_out > 0) {
            int r = inflate(&zs, Z_NO_FLUSH);
            if (r == Z_STREAM_ERROR) throw std::runtime_error("inflate(): stream error!");
            if (r == Z_NEED_DICT) throw std::runtime_error("inflate(): need dictionary!");
            if (r == Z_DATA_ERROR) throw std::runtime_error("inflate(): data error!");
            if (r == Z_MEM_ERROR) throw std::runtime_error("inflate(): memory error!");
            if (zs.avail_out > 0 && r == Z_STREAM_END) throw std::runtime_error("inflate(): attempting to read past the end of the stream!");
            if (zs.avail_out > 0 && zs.avail_in == 0 && r == Z_BUF_ERROR) throw std::runtime_error("Read less data than expected");
        }
    }
};
}  // namespace

static float UnitAngle(V3 u, V3 v) {  // loadserialized.cpp:95-102
    if (dot(u, v) < 0)
        return (c_PI - 2.0f) * std::asin(0.5f * length(v + u));
    else
        return 2.0f * std::asin(0.5f * length(v - u));
}

static void ComputeNormal(const std::vector<V3> &P, const std::vector<uint32_t> &idx, std::vector<V3> &N, bool flipNormals) {
    N.assign(P.size(), V3{0, 0, 0});  // Nelson Max weights, loadserialized.cpp:104-151
    for (size_t t = 0; t + 2 < idx.size(); t += 3) {
        V3 n{0, 0, 0};
        for (int i = 0; i < 3; ++i) {
            const V3 &v0 = P[idx[t + i]], &v1 = P[idx[t + (i + 1) % 3]], &v2 = P[idx[t + (i + 2) % 3]];
            V3 sideA = v1 - v0, sideB = v2 - v0;
            if (i == 0) {
                n = cross(sideA, sideB);
                float len = length(n);
                if (len == 0) break;
                n = n * (1.0f / len);
            }
            float angle = UnitAngle(normalize(sideA), normalize(sideB));
            N[idx[t + i]] = N[idx[t + i]] + n * angle;
            if (flipNormals) N[idx[t + i]] = -N[idx[t + i]];
        }
    }
    for (auto &n : N) {
        float len = length(n);
        n = len != 0 ? n * (1.0f / len) : V3{0, 0, 0};
    }
}

static void LoadSerialized(const std::string &fn, int shapeIndex, const M4 &toWorld, bool flipNormals, bool faceNormals, Mesh &mesh) {
    std::ifstream fs(fn, std::ios::binary);
    if (!fs) throw std::runtime_error("File not found: " + fn);
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(fs)), std::istreambuf_iterator<char>());
    if (buf.size() < 8) throw std::runtime_error("Read less data than expected");
    uint16_t version;
    memcpy(&version, &buf[2], 2);
    size_t offset = 0;
    if (shapeIndex > 0) {  // SkipToIdx, loadserialized.cpp:153-171
        uint32_t count;
        memcpy(&count, &buf[buf.size() - 4], 4);
        if ((uint32_t)shapeIndex >= count) throw std::runtime_error("shapeIndex out of range");
        if (version == 4) {
            uint64_t o;
            memcpy(&o, &buf[buf.size() - 8 * (count - shapeIndex) - 4], 8);
            offset = (size_t)o;
        } else {
            uint32_t o;
            memcpy(&o, &buf[buf.size() - 4 * (count - shapeIndex + 1)], 4);
            offset = o;
        }
    }
    ZReader zs(&buf[offset + 4], buf.size() - offset - 4);
    uint32_t flags;
    zs.read(&flags, 4);
    if (version == 4) {
        char c;
        do zs.read(&c, 1);
        while (c != '\0');
    }
    uint64_t nv = 0, nt = 0;
    zs.read(&nv, 8);
    zs.read(&nt, 8);
    const bool dbl = flags & 0x2000;
    faceNormals = (flags & 0x0010) || faceNormals;
    M4 inv = Inverse(toWorld);
    auto rd = [&](int n, float *out) {
        if (dbl) {
            double d[3];
            zs.read(d, 8 * n);
            for (int i = 0; i < n; i++) out[i] = (float)d[i];
        } else
            zs.read(out, 4 * n);
    };
    mesh.P.resize(nv);
    for (size_t i = 0; i < nv; i++) {
        float p[3];
        rd(3, p);
        mesh.P[i] = XformPoint(toWorld, V3{p[0], p[1], p[2]});
    }
    if (flags & 0x0001) {
        mesh.N.resize(nv);
        for (size_t i = 0; i < nv; i++) {
            float p[3];
            rd(3, p);
            mesh.N[i] = XformNormal(inv, V3{p[0], p[1], p[2]});
            if (flipNormals) mesh.N[i] = -mesh.N[i];
        }
    }
    if (flags & 0x0002) {
        mesh.ST.resize(nv);
        for (size_t i = 0; i < nv; i++) {
            float p[2];
            rd(2, p);
            mesh.ST[i] = V2{p[0], p[1]};
        }
    }
    if (flags & 0x0008) {
        std::vector<double> col(nv * 3);
        zs.read(col.data(), col.size() * 8);
    }
    mesh.idx.resize(nt * 3);
    zs.read(mesh.idx.data(), nt * 12);
    if (mesh.N.empty() || faceNormals) ComputeNormal(mesh.P, mesh.idx, mesh.N, flipNormals);
}

// parseobj.cpp:57-275 (v / vt / vn / f with v, v/vt, v//vn, v/vt/vn; quads split (0,1,2),(0,2,3))
static void LoadObj(const std::string &fn, const M4 &toWorld, bool flipNormals, bool faceNormals, Mesh &mesh) {
    std::ifstream ifs(fn);
    if (!ifs) throw std::runtime_error("File not found: " + fn);
    M4 inv = Inverse(toWorld);
    std::vector<V3> pool_p, pool_n;
    std::vector<V2> pool_st;
    std::map<std::array<int, 3>, uint32_t> vmap;
    auto getVertex = [&](int vi, int ti, int ni) -> uint32_t {
        std::array<int, 3> key{vi, ti, ni};
        auto it = vmap.find(key);
        if (it != vmap.end()) return it->second;
        uint32_t id = (uint32_t)mesh.P.size();
        mesh.P.push_back(pool_p.at(vi - 1));
        if (ti != 0) mesh.ST.push_back(pool_st.at(ti - 1));
        if (ni != 0) mesh.N.push_back(pool_n.at(ni - 1));
        vmap[key] = id;
        return id;
    };
    std::string line;
    while (std::getline(ifs, line)) {
        size_t b = line.find_first_not_of(" \t\r\n");
        if (b == std::string::npos) continue;
        line = line.substr(b);
        std::stringstream ss(line);
        std::string token;
        ss >> token;
        if (token == "v") {
            float x, y, z, w = 1.f;
            ss >> x >> y >> z >> w;
            pool_p.push_back(V3{x / w, y / w, z / w});
        } else if (token == "vt") {
            float s, t, w;
            ss >> s >> t >> w;
            pool_st.push_back(V2{s, 1.f - t});
        } else if (token == "vn") {
            float x, y, z;
            ss >> x >> y >> z;
            pool_n.push_back(normalize(V3{x, y, z}));
        } else if (token == "f") {
            std::string is;
            std::vector<uint32_t> ids;
            while (ss >> is) {
                int vi = 0, ti = 0, ni = 0;
                size_t s1 = is.find('/');
                if (s1 == std::string::npos)
                    vi = std::stoi(is);
                else {
                    vi = std::stoi(is.substr(0, s1));
                    size_t s2 = is.find('/', s1 + 1);
                    if (s2 == std::string::npos) {
                        if (s1 + 1 < is.size()) ti = std::stoi(is.substr(s1 + 1));
                    } else {
                        if (s2 > s1 + 1) ti = std::stoi(is.substr(s1 + 1, s2 - s1 - 1));
                        if (s2 + 1 < is.size()) ni = std::stoi(is.substr(s2 + 1));
                    }
                }
                if (vi < 0 || ti < 0 || ni < 0) throw std::runtime_error("Negative vertex reference");
                ids.push_back(getVertex(vi, ti, ni));
            }
            if (ids.size() < 3 || ids.size() > 4) throw std::runtime_error("Only triangles and quads are supported");
            mesh.idx.insert(mesh.idx.end(), {ids[0], ids[1], ids[2]});
            if (ids.size() == 4) mesh.idx.insert(mesh.idx.end(), {ids[0], ids[2], ids[3]});
        }
    }
    if (mesh.ST.size() != mesh.P.size()) mesh.ST.clear();
    bool hasN = mesh.N.size() == mesh.P.size();
    for (auto &p : mesh.P) p = XformPoint(toWorld, p);
    if (hasN && !faceNormals) {
        for (auto &n : mesh.N) {
            n = XformNormal(inv, n);
            if (flipNormals) n = -n;
        }
    } else
        ComputeNormal(mesh.P, mesh.idx, mesh.N, flipNormals);
}

// -------------------------------------------------------------------------------- materials
namespace {
struct ParseCtx {
    Scene *scene;
    std::string baseDir;
    std::map<std::string, int> bsdfMap;         // id -> material index
    std::map<std::string, TextureRef> texMap;   // id -> texture
    LoadOverrides ov;
};
}  // namespace

static float FastPow(float x, float p);

static TextureRef ParseTexture(ParseCtx &cx, const XmlNode &node) {
    if (node.attr("type") != "bitmap") throw std::runtime_error("Unknown texture type");
    std::string filename;
    float sc = 1.f;
    for (auto &c : node.children) {
        std::string name = c->attr("name");
        if (name == "filename") filename = c->attr("value");
        else if (name == "uvscale") sc = std::stof(c->attr("value"));
    }
    Bitmap bm;
    bm.filename = filename;
    bool is8 = false;
    bm.img = ReadImage(cx.baseDir + filename, &is8);
    bm.gamma = is8 ? 2.2f : 1.0f;
    double acc[3] = {0, 0, 0};
    for (size_t i = 0; i < bm.img.data.size(); i += 3)
        for (int k = 0; k < 3; k++) acc[k] += std::pow((double)bm.img.data[i + k], (double)bm.gamma);
    size_t npx = (size_t)bm.img.width * bm.img.height;
    for (int k = 0; k < 3; k++) bm.avg[k] = (float)(acc[k] / (double)npx);
    cx.scene->bitmaps.push_back(std::move(bm));
    TextureRef t;
    t.bitmap = (int)cx.scene->bitmaps.size() - 1;
    t.sScale = t.tScale = sc;
    return t;
}

static TextureRef ParseNDMap(ParseCtx &cx, const XmlNode &node, int N) {
    TextureRef t;
    if (node.name == "texture")
        t = ParseTexture(cx, node);
    else if (node.name == "ref") {
        auto it = cx.texMap.find(node.attr("id"));
        if (!node.has("id") || it == cx.texMap.end()) throw std::runtime_error("ref not found");
        t = it->second;
    } else {
        if (N == 1) {
            float v = std::stof(node.attr("value"));
            t.value[0] = t.value[1] = t.value[2] = v;
        } else {
            V3 v = ParseVector3(node.attr("value"));
            t.value[0] = v.x, t.value[1] = v.y, t.value[2] = v.z;
        }
    }
    return t;
}

static void TexAvg(const Scene &s, const TextureRef &t, float out[3]) {
    if (t.bitmap < 0) memcpy(out, t.value, 12);
    else memcpy(out, s.bitmaps[t.bitmap].avg, 12);
}

static TextureRef ConstTex(float a, float b, float c) {
    TextureRef t;
    t.value[0] = a, t.value[1] = b, t.value[2] = c;
    return t;
}

static int ParseBSDF(ParseCtx &cx, const XmlNode &node, bool twoSided = false) {
    std::string type = node.attr("type");
    Material m;
    m.twoSided = twoSided;
    if (type == "diffuse") {
        m.type = BSDF_LAMBERTIAN;
        m.Kd = ConstTex(0.5f, 0.5f, 0.5f);
        for (auto &c : node.children)
            if (c->attr("name") == "reflectance") m.Kd = ParseNDMap(cx, *c, 3);
    } else if (type == "phong") {
        m.type = BSDF_PHONG;
        m.Kd = ConstTex(0.5f, 0.5f, 0.5f);
        m.Ks = ConstTex(0.2f, 0.2f, 0.2f);
        m.expOrAlpha = ConstTex(30.f, 30.f, 30.f);
        for (auto &c : node.children) {
            std::string name = c->attr("name");
            if (name == "diffuseReflectance") m.Kd = ParseNDMap(cx, *c, 3);
            else if (name == "specularReflectance") m.Ks = ParseNDMap(cx, *c, 3);
            else if (name == "exponent") m.expOrAlpha = ParseNDMap(cx, *c, 1);
        }
        float ks[3], kd[3];
        TexAvg(*cx.scene, m.Ks, ks);
        TexAvg(*cx.scene, m.Kd, kd);
        float ksAvg = Lum(ks), kdAvg = Lum(kd), sum = ksAvg + kdAvg;
        m.KsWeight = sum > 0.f ? ksAvg / sum : 0.f;
    } else if (type == "roughdielectric") {
        m.type = BSDF_ROUGHDIELECTRIC;
        m.Ks = ConstTex(1, 1, 1);
        m.Kt = ConstTex(1, 1, 1);
        float intIOR = 1.5046f, extIOR = 1.000277f;
        m.expOrAlpha = ConstTex(0.1f, 0.1f, 0.1f);
        for (auto &c : node.children) {
            std::string name = c->attr("name");
            if (name == "intIOR") intIOR = std::stof(c->attr("value"));
            else if (name == "extIOR") extIOR = std::stof(c->attr("value"));
            else if (name == "alpha") m.expOrAlpha = ParseNDMap(cx, *c, 1);
            else if (name == "specularReflectance") m.Ks = ParseNDMap(cx, *c, 3);
            else if (name == "specularTransmittance") m.Kt = ParseNDMap(cx, *c, 3);
        }
        m.eta = intIOR / extIOR;
        m.invEta = 1.0f / m.eta;
    } else if (type == "twosided") {
        for (auto &c : node.children)
            if (c->name == "bsdf") return ParseBSDF(cx, *c, true);
        throw std::runtime_error("Unknown BSDF");
    } else {
        printf("BSDF: %s not found.\n", type.c_str());
        throw std::runtime_error("Unknown BSDF");
    }
    if (cx.ov.forceDiffuse && m.type != BSDF_LAMBERTIAN) {
        // BASELINE.json config 2 ("Lambertian-only BSDF"): keep a constant, non-black diffuse
        // reflectance from the XML if there is one, else 0.5.
        Material d;
        d.type = BSDF_LAMBERTIAN;
        d.twoSided = m.twoSided;
        d.Kd = ConstTex(0.5f, 0.5f, 0.5f);
        if (m.type == BSDF_PHONG && m.Kd.bitmap < 0 && (m.Kd.value[0] + m.Kd.value[1] + m.Kd.value[2]) > 0.f) d.Kd = m.Kd;
        m = d;
    } else if (cx.ov.forceDiffuse && m.Kd.bitmap >= 0) {
        m.Kd = ConstTex(0.5f, 0.5f, 0.5f);
    }
    cx.scene->materials.push_back(m);
    return (int)cx.scene->materials.size() - 1;
}

// -------------------------------------------------------------------------------- shapes / emitters
static void FinishMesh(Mesh &mesh) {
    mesh.bmin = V3{INFINITY, INFINITY, INFINITY};
    mesh.bmax = V3{-INFINITY, -INFINITY, -INFINITY};
    for (auto &p : mesh.P)
        for (int k = 0; k < 3; k++) {
            mesh.bmin[k] = std::min(mesh.bmin[k], p[k]);
            mesh.bmax[k] = std::max(mesh.bmax[k], p[k]);
        }
    for (auto i : mesh.idx)
        if (i >= mesh.P.size()) throw std::runtime_error("mesh index out of range");
}

static void SetAreaLight(Mesh &mesh) {  // trianglemesh.cpp:269-285
    size_t nt = mesh.numTris();
    std::vector<float> area(nt);
    mesh.totalArea = 0.f;
    for (size_t i = 0; i < nt; i++) {
        V3 p0 = mesh.P[mesh.idx[3 * i]], p1 = mesh.P[mesh.idx[3 * i + 1]], p2 = mesh.P[mesh.idx[3 * i + 2]];
        area[i] = 0.5f * length(cross(p1 - p0, p2 - p0));
        mesh.totalArea += area[i];
    }
    BuildPiecewise1D(area.data(), (int)nt, mesh.areaFunc, mesh.areaCdf, mesh.areaFuncInt);
}

static void ParseShape(ParseCtx &cx, const XmlNode &node) {
    Scene &S = *cx.scene;
    int material = -1;
    for (auto &c : node.children) {
        if (c->name == "bsdf") {
            material = ParseBSDF(cx, *c);
            break;
        } else if (c->name == "ref") {
            auto it = cx.bsdfMap.find(c->attr("id"));
            if (!c->has("id") || it == cx.bsdfMap.end()) throw std::runtime_error("ref not found");
            material = it->second;
            break;
        }
    }
    std::string type = node.attr("type");
    if (type != "serialized" && type != "obj") {
        printf("shape type: %s not found.\n", type.c_str());
        throw std::runtime_error("Invalid shape");
    }
    std::string filename;
    int shapeIndex = 0;
    M4 toWorld = M4::identity();
    bool flipNormals = false, faceNormals = false;
    for (auto &c : node.children) {
        std::string name = c->attr("name");
        if (name == "filename") filename = c->attr("value");
        else if (name == "shapeIndex") shapeIndex = atoi(c->attr("value").c_str());
        else if (name == "flipNormals") flipNormals = true;  // reference quirk: true whenever present (parsescene.cpp:254)
        else if (name == "faceNormals") faceNormals = true;
        else if (name == "toWorld") {
            if (c->name == "transform") toWorld = ParseTransform(*c);
            else if (c->name == "animation") throw std::runtime_error("moving geometry is not supported by the MI355X back end (SURVEY.md §8: static scenes)");
        }
    }
    Mesh mesh;
    mesh.material = material;
    if (type == "serialized") LoadSerialized(cx.baseDir + filename, shapeIndex, toWorld, flipNormals, faceNormals, mesh);
    else LoadObj(cx.baseDir + filename, toWorld, flipNormals, faceNormals, mesh);
    FinishMesh(mesh);
    if (material < 0) throw std::runtime_error("Invalid shape");  // reference dereferences a null bsdf later
    for (auto &c : node.children)
        if (c->name == "emitter") {
            V3 radiance{1, 1, 1};
            for (auto &g : c->children)
                if (g->attr("name") == "radiance") radiance = ParseVector3(g->attr("value"));
            Light L;
            L.type = LIGHT_AREA;
            L.samplingWeight = 1.f;
            L.mesh = (int)S.meshes.size();
            L.radiance = radiance;
            SetAreaLight(mesh);
            mesh.areaLight = (int)S.lights.size();
            S.lights.push_back(std::move(L));
        }
    S.meshes.push_back(std::move(mesh));
}

static void CreateEnvmapSampleInfo(Light &L) {  // envlight.cpp:24-63
    const Image3f &image = L.image;
    int height = image.height, width = image.width;
    EnvmapSampleInfo &si = L.sampleInfo;
    si.cdfCols.assign((size_t)(width + 1) * height, 0.f);
    si.cdfRows.assign(height + 1, 0.f);
    si.rowWeights.assign(height, 0.f);
    size_t colPos = 0, rowPos = 0;
    float rowSum = 0.f;
    si.cdfRows[rowPos++] = 0.f;
    for (int y = 0; y < height; y++) {
        float colSum = 0.f;
        si.cdfCols[colPos++] = 0.f;
        for (int x = 0; x < width; x++) {
            colSum += Lum(image.At(x, y));
            si.cdfCols[colPos++] = colSum;
        }
        float normalization = 1.0f / colSum;
        for (int x = 1; x < width; x++) si.cdfCols[colPos - x - 1] *= normalization;
        si.cdfCols[colPos - 1] = 1.f;
        float weight = std::sin((y + 0.5f) * c_PI / (float)height);
        si.rowWeights[y] = weight;
        rowSum += colSum * weight;
        si.cdfRows[rowPos++] = rowSum;
    }
    float normalization = 1.0f / rowSum;
    for (int y = 1; y < height; y++) si.cdfRows[rowPos - y - 1] *= normalization;
    si.cdfRows[rowPos - 1] = 1.f;
    if (rowSum == 0 || !std::isfinite(rowSum)) throw std::runtime_error("Invalid environment map");
    si.normalization = 1.0f / (rowSum * (c_TWOPI / width) * (c_PI / height));
    si.pixelSize[0] = c_TWOPI / width;
    si.pixelSize[1] = (float)(M_PI / height);
}

static void ParseEmitter(ParseCtx &cx, const XmlNode &node) {
    Scene &S = *cx.scene;
    std::string type = node.attr("type");
    Light L;
    if (type == "point") {
        L.type = LIGHT_POINT;
        for (auto &c : node.children) {
            std::string name = c->attr("name");
            if (name == "position") {
                auto f = [&](const char *k) { return c->has(k) ? std::stof(c->attr(k)) : 0.f; };
                L.position = V3{f("x"), f("y"), f("z")};
            } else if (name == "intensity")
                L.intensity = ParseVector3(c->attr("value"));
        }
    } else if (type == "envmap") {
        L.type = LIGHT_ENV;
        std::string filename;
        L.toWorld = MakeAnimXform(M4::identity(), M4::identity());
        for (auto &c : node.children) {
            std::string name = c->attr("name");
            if (name == "filename") filename = c->attr("value");
            else if (name == "toWorld") {
                if (c->name == "transform") {
                    M4 m = ParseTransform(*c);
                    L.toWorld = MakeAnimXform(m, m);
                } else if (c->name == "animation")
                    L.toWorld = ParseAnimatedTransform(*c);
            }
        }
        L.toLight = Invert(L.toWorld);
        L.image = ReadImage(cx.baseDir + filename);
        CreateEnvmapSampleInfo(L);
        S.envLight = (int)S.lights.size();
    } else
        throw std::runtime_error("Unsupported emitter");
    S.lights.push_back(std::move(L));
}

static void ParseSensor(ParseCtx &cx, const XmlNode &node) {
    Scene &S = *cx.scene;
    Camera &cam = S.camera;
    cam.camToWorld = MakeAnimXform(M4::identity(), M4::identity());
    for (auto &c : node.children) {
        std::string name = c->attr("name");
        if (name == "nearClip") cam.nearClip = std::stof(c->attr("value"));
        else if (name == "farClip") cam.farClip = std::stof(c->attr("value"));
        else if (name == "fov") cam.fov = std::stof(c->attr("value"));
        else if (name == "toWorld") {
            if (c->name == "transform") {
                M4 m = ParseTransform(*c);
                cam.camToWorld = MakeAnimXform(m, m);
            } else if (c->name == "animation")
                cam.camToWorld = ParseAnimatedTransform(*c);
        } else if (c->name == "film") {
            for (auto &g : c->children) {
                std::string gn = g->attr("name");
                if (gn == "width") cam.width = atoi(g->attr("value").c_str());
                else if (gn == "height") cam.height = atoi(g->attr("value").c_str());
                else if (gn == "filename") S.outputName = g->attr("value");
            }
        }
    }
}

static void FinishCamera(Camera &cam) {  // camera.cpp:12-28
    cam.worldToCamera = Invert(cam.camToWorld);
    float aspect = (float)cam.width / (float)cam.height;
    cam.camToSample = Scale(V3{-0.5f, -0.5f * aspect, 1.0f}) * Translate(V3{-1.0f, -1.0f / aspect, 0.0f}) * Perspective(cam.fov, cam.nearClip, cam.farClip);
    cam.sampleToCam = Inverse(cam.camToSample);
    cam.dist = cam.width / (2.0f * (float)std::tan((double)((cam.fov / 2.0f) * (c_PI / 180.0f))));
}

static void ParseDptOptions(DptOptions &o, const XmlNode &node) {  // parsescene.cpp:535-590
    for (auto &c : node.children) {
        std::string name = c->attr("name"), v = c->attr("value");
        if (name == "integrator") o.integrator = v;
        else if (name == "spp") o.spp = std::stoi(v);
        else if (name == "bidirectional") o.bidirectional = v == "true";
        else if (name == "numinitsamples") o.numInitSamples = std::stoi(v);
        else if (name == "largestepprob") o.largeStepProbability = std::stof(v);
        else if (name == "largestepscale") o.largeStepProbScale = std::stof(v);
        else if (name == "mindepth") o.minDepth = std::stoi(v);
        else if (name == "maxdepth") o.maxDepth = std::stoi(v);
        else if (name == "directspp") o.directSpp = std::stoi(v);
        else if (name == "perturbstddev") o.perturbStdDev = std::stof(v);
        else if (name == "roughnessthreshold") o.roughnessThreshold = std::stof(v);
        else if (name == "uniformmixprob") o.uniformMixingProbability = std::stof(v);
        else if (name == "numchains") o.numChains = std::stoi(v);
        else if (name == "seedoffset") o.seedOffset = std::stoi(v);
        else if (name == "reportintervalspp") o.reportIntervalSpp = std::stoi(v);
        else if (name == "uselightcoordinatesampling") o.useLightCoordinateSampling = v == "true";
        else if (name == "largestepmultiplexed") o.largeStepMultiplexed = v == "true";
        else if (name == "h2mc") o.h2mc = v == "true";
        else if (name == "mala") o.mala = v == "true";
        else if (name == "mala-stepsize") o.malaStepsize = std::stof(v);
        else if (name == "mala-gn") o.malaGN = std::stof(v);
        else if (name == "samplecache") o.sampleFromGlobalCache = v == "true";
        else std::cerr << "Unknown dpt option:" << name << std::endl;
    }
}

std::unique_ptr<Scene> ParseSceneString(const std::string &xml, const std::string &baseDir, const LoadOverrides &ov) {
    XmlParser parser(xml);
    std::unique_ptr<XmlNode> doc = parser.parse();
    const XmlNode *root = doc->child("scene");
    if (!root) throw std::runtime_error("Parse error");
    std::unique_ptr<Scene> scene(new Scene);
    ParseCtx cx{scene.get(), baseDir, {}, {}, ov};
    bool haveCamera = false;
    for (auto &cp : root->children) {
        const XmlNode &child = *cp;
        if (child.name == "sensor") {
            ParseSensor(cx, child);
            haveCamera = true;
        } else if (child.name == "shape")
            ParseShape(cx, child);
        else if (child.name == "bsdf")
            cx.bsdfMap[child.attr("id")] = ParseBSDF(cx, child);
        else if (child.name == "emitter")
            ParseEmitter(cx, child);
        else if (child.name == "texture")
            cx.texMap[child.attr("id")] = ParseTexture(cx, child);
        else if (child.name == "dpt")
            ParseDptOptions(scene->options, child);
    }
    if (!haveCamera) throw std::runtime_error("scene has no sensor");
    Scene &S = *scene;
    if (ov.maxDepth > 0) S.options.maxDepth = ov.maxDepth;
    if (ov.numChains > 0) S.options.numChains = ov.numChains;
    if (ov.spp > 0) S.options.spp = ov.spp;
    if (ov.numInitSamples > 0) S.options.numInitSamples = ov.numInitSamples;
    if (ov.seedOffset >= 0) S.options.seedOffset = ov.seedOffset;
    if (ov.directSpp >= 0) S.options.directSpp = ov.directSpp;
    if (ov.width > 0) S.camera.width = ov.width;
    if (ov.height > 0) S.camera.height = ov.height;
    FinishCamera(S.camera);
    // light cdf (scene.cpp:21-28)
    if (S.lights.empty()) throw std::runtime_error("scene has no emitter");
    std::vector<float> w(S.lights.size());
    S.lightWeightSum = 0.f;
    for (size_t i = 0; i < w.size(); i++) {
        w[i] = S.lights[i].samplingWeight;
        S.lightWeightSum += w[i];
    }
    BuildPiecewise1D(w.data(), (int)w.size(), S.lightFunc, S.lightCdf, S.lightFuncInt);
    // bounding sphere (bounds.h:36-46, scene.cpp:33-40)
    V3 bmin{INFINITY, INFINITY, INFINITY}, bmax{-INFINITY, -INFINITY, -INFINITY};
    for (auto &m : S.meshes)
        for (int k = 0; k < 3; k++) {
            bmin[k] = std::min(bmin[k], m.bmin[k]);
            bmax[k] = std::max(bmax[k], m.bmax[k]);
        }
    S.bsphereCenter = 0.5f * (bmin + bmax);
    S.bsphereRadius = 0.5f * length(bmax - bmin);
    S.bsphereRadius *= 1000.0f;
    return scene;
}

std::unique_ptr<Scene> ParseScene(const std::string &filename, const LoadOverrides &ov) {
    std::ifstream f(filename);
    if (!f) throw std::runtime_error("Parse error: cannot open " + filename);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string dir;
    size_t sl = filename.rfind('/');
    if (sl != std::string::npos) dir = filename.substr(0, sl + 1);
    return ParseSceneString(ss.str(), dir, ov);
}

void SerializeSceneBlock(const Scene &scene, float out[38]) {
    float *b = out;
    *b++ = scene.options.useLightCoordinateSampling ? 1.f : 0.f;
    for (int i = 0; i < 4; i++)  // column-major (utils.h:331-339)
        for (int j = 0; j < 4; j++) *b++ = scene.camera.sampleToCam.m[j][i];
    const AnimXform &x = scene.camera.camToWorld;
    *b++ = x.isMoving;
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < 3; i++) *b++ = x.t[k][i];
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < 4; i++) *b++ = x.q[k][i];
    *b++ = float(scene.camera.height * scene.camera.width);
    *b++ = scene.camera.dist;
    *b++ = scene.bsphereCenter.x, *b++ = scene.bsphereCenter.y, *b++ = scene.bsphereCenter.z;
    *b++ = scene.bsphereRadius;
}

// fastpow from the reference's fastmath.h (Mineiro's fastapprox: fastpow2(p * fastlog2(x))), used by
// bitmaptexture.h:94.  Restated from the published formulas.
static float FastLog2(float x) {
    union { float f; uint32_t i; } vx = {x};
    union { uint32_t i; float f; } mx = {(vx.i & 0x007FFFFF) | 0x3f000000};
    float y = (float)vx.i;
    y *= 1.1920928955078125e-7f;
    return y - 124.22551499f - 1.498030302f * mx.f - 1.72587999f / (0.3520887068f + mx.f);
}
static float FastPow2(float p) {
    float offset = (p < 0) ? 1.0f : 0.0f;
    float clipp = (p < -126) ? -126.0f : p;
    int w = (int)clipp;
    float z = clipp - w + offset;
    union { uint32_t i; float f; } v = {(uint32_t)((1 << 23) * (clipp + 121.2740575f + 27.7280233f / (4.84252568f - z) - 1.49012907f * z))};
    return v.f;
}
static float FastPow(float x, float p) { return FastPow2(p * FastLog2(x)); }

void EvalTexture(const Scene &scene, const TextureRef &t, float s, float tt, float out[3]) {
    if (t.bitmap < 0) {
        memcpy(out, t.value, 12);
        return;
    }
    // Periodic bilinear lookup standing in for OIIO's TextureSystem::texture() with zero filter width
    // (bitmaptexture.h:72-97).  OIIO is third-party and unbuilt here: parity unpinned.
    const Bitmap &bm = scene.bitmaps[t.bitmap];
    const int W = bm.img.width, H = bm.img.height;
    float fs = t.sScale * s * W - 0.5f, ft = t.tScale * tt * H - 0.5f;
    float x0f = std::floor(fs), y0f = std::floor(ft);
    float dx = fs - x0f, dy = ft - y0f;
    auto wrap = [](long v, int n) { long r = v % n; return (int)(r < 0 ? r + n : r); };
    int x0 = wrap((long)x0f, W), x1 = wrap((long)x0f + 1, W), y0 = wrap((long)y0f, H), y1 = wrap((long)y0f + 1, H);
    for (int k = 0; k < 3; k++) {
        float v = (1 - dx) * (1 - dy) * bm.img.At(x0, y0)[k] + dx * (1 - dy) * bm.img.At(x1, y0)[k] + (1 - dx) * dy * bm.img.At(x0, y1)[k] + dx * dy * bm.img.At(x1, y1)[k];
        out[k] = FastPow(std::max(v, 0.f), bm.gamma);
    }
}

}  // namespace lmc
