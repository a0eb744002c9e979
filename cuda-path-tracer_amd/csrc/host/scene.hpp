// Host-side scene description: the same classes and method names a user of the reference sees
// (src/scene/core.hpp:116-240, src/scene/mesh.hpp, src/scene/material.hpp, src/scene/camera.hpp,
// src/scene/bbox.hpp), backed by plain std::vector storage.  Nothing here touches the GPU: device
// tables are produced by PathTracer::BuildBuffers, so scenes can be loaded and inspected on a
// machine without a device.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <string_view>
#include <type_traits>
#include <vector>

#include "../vecmath.h"
#include "image.hpp"

struct Bbox {                                     // src/scene/bbox.hpp
    cpt::V3 pmin{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    cpt::V3 pmax{-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    cpt::V3 Centroid() const;
    float Extent() const;
    void Empty();
    Bbox TransformBy(const cpt::M4 &mat) const;
    void Merge(const Bbox &rhs);
    void Merge(const cpt::V3 &p);
    bool IntersectWith(const Bbox &rhs) const;
};

class Mesh {                                      // src/scene/mesh.hpp:11-55
public:
    void SetPositions(std::vector<cpt::V3> &&positions);
    void SetNormals(std::vector<cpt::V3> &&normals) { normals_ = std::move(normals); }
    void SetTexcoords(std::vector<cpt::V2> &&texcoords) { texcoords_ = std::move(texcoords); }
    void SetIndices(std::vector<uint32_t> &&indices) { indices_ = std::move(indices); }
    void CalcNormals();
    void CalcTexcoords();

    size_t VericesCount() const { return positions_.size(); }
    const std::vector<cpt::V3> &Positions() const { return positions_; }
    const std::vector<cpt::V3> &Normals() const { return normals_; }
    const std::vector<cpt::V2> &Texcoords() const { return texcoords_; }
    size_t IndicesCount() const { return indices_.size(); }
    const std::vector<uint32_t> &Indices() const { return indices_; }
    const struct Bbox &Bbox() const { return bbox_; }

private:
    std::vector<cpt::V3> positions_;
    std::vector<cpt::V3> normals_;
    std::vector<cpt::V2> texcoords_;
    std::vector<uint32_t> indices_;
    struct Bbox bbox_;
};

class MeshComponent {
public:
    void SetMesh(std::shared_ptr<Mesh> mesh) { mesh_ = std::move(mesh); }
    Mesh *GetMesh() const { return mesh_.get(); }
    const std::shared_ptr<Mesh> &Shared() const { return mesh_; }
private:
    std::shared_ptr<Mesh> mesh_;
};

class Material {                                  // src/scene/material.hpp:12-43
public:
    bool IsEmissive() const;
    void SetChanged() { ++revision; }

    std::string name;
    cpt::V3 emission{0.0f, 0.0f, 0.0f};
    cpt::V3 diffuse{0.5f, 0.5f, 0.5f};
    cpt::V3 specular{0.5f, 0.5f, 0.5f};
    cpt::V3 transmittance{0.0f, 0.0f, 0.0f};
    float ior = 1.5f;
    float shininess = 1.0f;
    float opacity = 1.0f;
    int bsdf_type = 2;
    // material maps: file names as the MTL gives them and the decoded images (null = constant);
    // the reference loads diffuse, specular and emissive maps (src/scene/loader.cpp:245-247)
    std::string emission_map, diffuse_map, specular_map;
    std::shared_ptr<Image> emission_image, diffuse_image, specular_image;
    uint32_t revision = 0;
};

class MaterialComponent {
public:
    void SetMaterial(std::shared_ptr<Material> material) { material_ = std::move(material); }
    Material *GetMaterial() const { return material_.get(); }
    const std::shared_ptr<Material> &Shared() const { return material_; }
private:
    std::shared_ptr<Material> material_;
};

class CameraComponent {                           // src/scene/camera.hpp:9-29
public:
    void Update() {}
    cpt::V3 pos{0.0f, 0.0f, -1.0f};
    cpt::V3 look_at{0.0f, 0.0f, 0.0f};
    cpt::V3 up{0.0f, 1.0f, 0.0f};
    float fov = 45.0f;
    uint32_t film_width = 1280;
    uint32_t film_height = 720;
};

class PathTracer;

// inverse of an object transform exactly as glm::inverse computes it (src/pathtracer/pathtracer.cpp:113)
cpt::M4 InverseTransform(const cpt::M4 &mat);

// One scene object with at most one component of each kind (the reference's block-list ECS,
// src/scene/core.hpp:13-110, is an implementation detail; iteration order = insertion order).
class SceneObject {
public:
    explicit SceneObject(std::string_view name) : name_(name) {}
    const std::string &Name() const { return name_; }

    template <typename T, typename... Args> T *AddComponent(Args &&...args);
    template <typename T> T *GetComponent() const;
    template <typename T> bool HasComponent() const { return GetComponent<T>() != nullptr; }

    cpt::M4 GetTransform() const;                 // T * Rz * Ry * Rx * S, angles in degrees (src/scene/core.cpp:109-117)

    cpt::V3 translate{0.0f, 0.0f, 0.0f};
    cpt::V3 rotate{0.0f, 0.0f, 0.0f};
    cpt::V3 scale{1.0f, 1.0f, 1.0f};

private:
    std::string name_;
    std::unique_ptr<MeshComponent> mesh_;
    std::unique_ptr<MaterialComponent> material_;
    std::unique_ptr<CameraComponent> camera_;
    std::shared_ptr<PathTracer> path_tracer_;
    friend class Scene;
};

class Scene {                                     // src/scene/core.hpp:116-173
public:
    std::shared_ptr<SceneObject> AddObject(std::string_view name);

    template <typename... Ts, typename F> void ForEach(F &&fn);
    template <typename... Ts> std::shared_ptr<SceneObject> FirstObjectWith() const;
    template <typename T> size_t Count() const;

    // one step of every component that has an Update(): cameras, then path tracers (one sample each)
    void Update();

    const std::vector<std::shared_ptr<SceneObject>> &Objects() const { return objects_; }

private:
    std::vector<std::shared_ptr<SceneObject>> objects_;
};

// ---- template bodies
template <typename T> T *SceneObject::GetComponent() const {
    using U = std::remove_const_t<T>;
    if constexpr (std::is_same_v<U, MeshComponent>) return mesh_.get();
    else if constexpr (std::is_same_v<U, MaterialComponent>) return material_.get();
    else if constexpr (std::is_same_v<U, CameraComponent>) return camera_.get();
    else if constexpr (std::is_same_v<U, PathTracer>) return path_tracer_.get();
    else return nullptr;
}

template <typename T, typename... Args> T *SceneObject::AddComponent(Args &&...args) {
    if constexpr (std::is_same_v<T, MeshComponent>) { mesh_ = std::make_unique<MeshComponent>(std::forward<Args>(args)...); return mesh_.get(); }
    else if constexpr (std::is_same_v<T, MaterialComponent>) { material_ = std::make_unique<MaterialComponent>(std::forward<Args>(args)...); return material_.get(); }
    else if constexpr (std::is_same_v<T, CameraComponent>) { camera_ = std::make_unique<CameraComponent>(std::forward<Args>(args)...); return camera_.get(); }
    else if constexpr (std::is_same_v<T, PathTracer>) { path_tracer_ = std::make_shared<PathTracer>(std::forward<Args>(args)...); return path_tracer_.get(); }
    else { static_assert(sizeof(T) == 0, "unknown component type"); return nullptr; }
}

template <typename... Ts, typename F> void Scene::ForEach(F &&fn) {
    for (auto &obj : objects_) {
        if ((obj->template HasComponent<Ts>() && ...)) {
            if constexpr (std::is_invocable_v<F, SceneObject &, Ts &...>) fn(*obj, *obj->template GetComponent<Ts>()...);
            else fn(*obj->template GetComponent<Ts>()...);
        }
    }
}

template <typename... Ts> std::shared_ptr<SceneObject> Scene::FirstObjectWith() const {
    for (auto &obj : objects_)
        if ((obj->template HasComponent<Ts>() && ...)) return obj;
    return nullptr;
}

template <typename T> size_t Scene::Count() const {
    size_t n = 0;
    for (auto &obj : objects_) n += obj->template HasComponent<T>() ? 1 : 0;
    return n;
}
