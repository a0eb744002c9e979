// Binary copy of a loaded .obj scene (LoadObjSceneCached in loader.hpp).  Little-endian, this
// program's own format:
//   "CPTSCN01", u32 bsdf_type, u32 number of source files, per source file {string path, u64 size, i64 mtime},
//   u32 number of materials, per material {string name, 4 x 3 floats emission/diffuse/specular/transmittance,
//   3 floats ior/shininess/opacity, 3 strings emission/diffuse/specular map}, u32 number of objects,
//   per object {string name, u32 material, u64 vertices, u64 indices, positions, normals, texcoords, indices}.
// A string is a u32 length and its bytes.
#include <cstdio>
#include <cstring>
#include <iostream>

#include "loader.hpp"

namespace {

namespace fs = std::filesystem;
constexpr char kMagic[8] = {'C', 'P', 'T', 'S', 'C', 'N', '0', '1'};

struct Source { std::string path; uint64_t size; int64_t mtime; };

bool Stat(const fs::path &p, Source &out) {
    std::error_code ec;
    const auto size = fs::file_size(p, ec);
    if (ec) return false;
    const auto time = fs::last_write_time(p, ec);
    if (ec) return false;
    out.path = p.string();
    out.size = (uint64_t)size;
    out.mtime = (int64_t)time.time_since_epoch().count();
    return true;
}

struct Writer {
    FILE *f;
    bool ok = true;
    void Bytes(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    template <typename T> void Pod(const T &v) { Bytes(&v, sizeof(T)); }
    void String(const std::string &s) { Pod((uint32_t)s.size()); Bytes(s.data(), s.size()); }
};

struct Reader {
    FILE *f;
    bool ok = true;
    void Bytes(void *p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; }
    template <typename T> T Pod() { T v{}; Bytes(&v, sizeof(T)); return v; }
    // bytes between the read position and the end of the file (0 when that cannot be told)
    uint64_t Remaining() {
        const long here = ftell(f);
        if (here < 0 || fseek(f, 0, SEEK_END) != 0) return 0;
        const long end = ftell(f);
        if (fseek(f, here, SEEK_SET) != 0 || end < here) { ok = false; return 0; }
        return (uint64_t)(end - here);
    }
    std::string String() {
        const uint32_t n = Pod<uint32_t>();
        if (!ok || n > (1u << 20)) { ok = false; return {}; }
        std::string s(n, '\0');
        Bytes(s.data(), n);
        return s;
    }
};

struct CachedObject {
    std::string name;
    uint32_t material;
    std::vector<cpt::V3> positions, normals;
    std::vector<cpt::V2> texcoords;
    std::vector<uint32_t> indices;
};

bool ReadCache(Scene &scene, const fs::path &cache, const fs::path &obj, int bsdf_type) {
    FILE *f = fopen(cache.c_str(), "rb");
    if (!f) return false;
    struct Close { FILE *f; ~Close() { fclose(f); } } close{f};
    Reader r{f};
    char magic[8];
    r.Bytes(magic, 8);
    if (!r.ok || memcmp(magic, kMagic, 8) != 0 || r.Pod<uint32_t>() != (uint32_t)bsdf_type) return false;
    const uint32_t num_sources = r.Pod<uint32_t>();
    if (!r.ok || num_sources == 0 || num_sources > 4096) return false;
    for (uint32_t i = 0; i < num_sources; i++) {
        Source was{r.String(), r.Pod<uint64_t>(), r.Pod<int64_t>()}, now;
        if (!r.ok) return false;
        if (i == 0 && was.path != obj.string()) return false;
        if (!Stat(was.path, now) || now.size != was.size || now.mtime != was.mtime) return false;
    }
    const uint32_t num_materials = r.Pod<uint32_t>();
    if (!r.ok || num_materials > (1u << 24)) return false;
    std::vector<std::shared_ptr<Material>> materials(num_materials);
    for (auto &m : materials) {
        m = std::make_shared<Material>();
        m->name = r.String();
        m->emission = r.Pod<cpt::V3>(); m->diffuse = r.Pod<cpt::V3>(); m->specular = r.Pod<cpt::V3>(); m->transmittance = r.Pod<cpt::V3>();
        m->ior = r.Pod<float>(); m->shininess = r.Pod<float>(); m->opacity = r.Pod<float>();
        m->bsdf_type = bsdf_type;
        m->emission_map = r.String(); m->diffuse_map = r.String(); m->specular_map = r.String();
        if (!r.ok) return false;
    }
    const uint32_t num_objects = r.Pod<uint32_t>();
    if (!r.ok || num_objects > (1u << 24)) return false;
    std::vector<CachedObject> objects(num_objects);
    for (auto &o : objects) {
        o.name = r.String();
        o.material = r.Pod<uint32_t>();
        const uint64_t nv = r.Pod<uint64_t>(), ni = r.Pod<uint64_t>();
        if (!r.ok || o.material >= num_materials || nv > 0xffffffffull || ni > (1ull << 40) || ni % 3 != 0) return false;
        // a damaged count must not size an allocation: the arrays it announces have to be in the file
        if (nv * (2 * sizeof(cpt::V3) + sizeof(cpt::V2)) + ni * sizeof(uint32_t) > r.Remaining()) return false;
        o.positions.resize(nv); o.normals.resize(nv); o.texcoords.resize(nv); o.indices.resize(ni);
        r.Bytes(o.positions.data(), nv * sizeof(cpt::V3));
        r.Bytes(o.normals.data(), nv * sizeof(cpt::V3));
        r.Bytes(o.texcoords.data(), nv * sizeof(cpt::V2));
        r.Bytes(o.indices.data(), ni * sizeof(uint32_t));
        if (!r.ok) return false;
        for (uint32_t idx : o.indices) if (idx >= nv) return false;
    }
    if (fgetc(f) != EOF) return false;      // trailing bytes: not a file this code wrote

    // everything checked: only now does the scene change
    const fs::path base_dir = obj.parent_path();
    for (auto &m : materials) {
        auto load_map = [&](std::shared_ptr<Image> &image, const std::string &name) {
            if (name.empty()) return;
            std::string error;
            image = LoadImage(base_dir / name, error);
            if (!image) std::cout << "warning: material '" << m->name << "': " << error << "; the constant is used" << std::endl;
        };
        load_map(m->diffuse_image, m->diffuse_map);
        load_map(m->specular_image, m->specular_map);
        load_map(m->emission_image, m->emission_map);
        m->SetChanged();
    }
    for (auto &o : objects) {
        auto mesh = std::make_shared<Mesh>();
        mesh->SetPositions(std::move(o.positions));
        mesh->SetNormals(std::move(o.normals));
        mesh->SetTexcoords(std::move(o.texcoords));
        mesh->SetIndices(std::move(o.indices));
        auto object = scene.AddObject(o.name);
        object->AddComponent<MeshComponent>()->SetMesh(mesh);
        object->AddComponent<MaterialComponent>()->SetMaterial(materials[o.material]);
    }
    return true;
}

bool WriteCache(const Scene &scene, size_t first_object, const fs::path &cache, const std::vector<Source> &sources, int bsdf_type) {
    const fs::path partial = cache.string() + ".partial";
    FILE *f = fopen(partial.c_str(), "wb");
    if (!f) return false;
    Writer w{f};
    w.Bytes(kMagic, 8);
    w.Pod((uint32_t)bsdf_type);
    w.Pod((uint32_t)sources.size());
    for (auto &s : sources) { w.String(s.path); w.Pod(s.size); w.Pod(s.mtime); }
    std::vector<const Material *> materials;
    std::vector<uint32_t> material_of;
    const auto &all = scene.Objects();
    for (size_t i = first_object; i < all.size(); i++) {
        const Material *m = all[i]->GetComponent<MaterialComponent>()->GetMaterial();
        size_t k = 0;
        while (k < materials.size() && materials[k] != m) k++;
        if (k == materials.size()) materials.push_back(m);
        material_of.push_back((uint32_t)k);
    }
    w.Pod((uint32_t)materials.size());
    for (const Material *m : materials) {
        w.String(m->name);
        w.Pod(m->emission); w.Pod(m->diffuse); w.Pod(m->specular); w.Pod(m->transmittance);
        w.Pod(m->ior); w.Pod(m->shininess); w.Pod(m->opacity);
        w.String(m->emission_map); w.String(m->diffuse_map); w.String(m->specular_map);
    }
    w.Pod((uint32_t)(all.size() - first_object));
    for (size_t i = first_object; i < all.size(); i++) {
        const Mesh *mesh = all[i]->GetComponent<MeshComponent>()->GetMesh();
        w.String(all[i]->Name());
        w.Pod(material_of[i - first_object]);
        w.Pod((uint64_t)mesh->VericesCount());
        w.Pod((uint64_t)mesh->IndicesCount());
        w.Bytes(mesh->Positions().data(), mesh->Positions().size() * sizeof(cpt::V3));
        w.Bytes(mesh->Normals().data(), mesh->Normals().size() * sizeof(cpt::V3));
        w.Bytes(mesh->Texcoords().data(), mesh->Texcoords().size() * sizeof(cpt::V2));
        w.Bytes(mesh->Indices().data(), mesh->Indices().size() * sizeof(uint32_t));
    }
    const bool ok = (fclose(f) == 0) && w.ok;
    std::error_code ec;
    if (ok) fs::rename(partial, cache, ec);
    if (!ok || ec) { fs::remove(partial, ec); return false; }
    return true;
}

}  // namespace

bool LoadObjSceneCached(Scene &scene, const std::filesystem::path &path, int bsdf_type, bool *from_cache) {
    const fs::path cache = path.string() + ".cptcache";
    if (from_cache) *from_cache = false;
    if (ReadCache(scene, cache, path, bsdf_type)) {
        if (from_cache) *from_cache = true;
        return true;
    }
    // the sources are described as they are before they are read: a file that changes meanwhile makes the
    // copy stale at once instead of wrong
    std::vector<Source> sources(1);
    const bool obj_known = Stat(path, sources[0]);
    const size_t first_object = scene.Objects().size();
    std::vector<fs::path> libraries;
    if (!LoadObjScene(scene, path, bsdf_type, &libraries)) return false;
    bool describable = obj_known;
    for (auto &lib : libraries) {
        Source s;
        if (Stat(lib, s)) sources.push_back(s);
        else describable = false;           // a library that is missing now may appear later: no copy
    }
    // every mesh must carry all three vertex arrays at full length for the fixed-size records
    for (size_t i = first_object; i < scene.Objects().size() && describable; i++) {
        const Mesh *mesh = scene.Objects()[i]->GetComponent<MeshComponent>()->GetMesh();
        describable = mesh->Normals().size() == mesh->VericesCount() && mesh->Texcoords().size() == mesh->VericesCount();
    }
    if (describable) WriteCache(scene, first_object, cache, sources, bsdf_type);
    return true;
}
