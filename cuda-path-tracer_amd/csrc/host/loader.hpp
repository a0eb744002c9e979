// Scene file readers with the reference's signatures (src/scene/loader.hpp:7-11).
#pragma once
#include <filesystem>

#include "scene.hpp"

// One SceneObject per (shape, material) of the .obj, each with a MeshComponent and a
// MaterialComponent; bsdf_type: 0 lambert, 1 phong, 2 blinn-phong, 3 microfacet.
bool LoadObjScene(Scene &scene, const std::filesystem::path &path, int bsdf_type);
// The same, also reporting the material libraries the .obj named (in the order it named them).
bool LoadObjScene(Scene &scene, const std::filesystem::path &path, int bsdf_type, std::vector<std::filesystem::path> *material_libraries);
// The same scene through a binary copy kept next to the .obj (<path>.cptcache): read when it was written from
// the .obj and material libraries as they are now (sizes and modification times) and for the same bsdf_type,
// otherwise the text is parsed and the copy (re)written.  Material maps are always read from their image files.
// A copy that cannot be written is not an error.  *from_cache tells which way it went.  The reference has no
// such file; a 10M-triangle .obj is about a gigabyte of text.
bool LoadObjSceneCached(Scene &scene, const std::filesystem::path &path, int bsdf_type, bool *from_cache);
// {"camera": {"pos", "look_at", "up", "fov", optional "resolution"}}
bool LoadJsonScene(Scene &scene, const std::filesystem::path &path);
// <camera width height fovy><eye/><lookat/><up/></camera> and <light mtlname radiance="r,g,b"/>
bool LoadXmlScene(Scene &scene, const std::filesystem::path &path);
