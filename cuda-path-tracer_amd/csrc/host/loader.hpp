// Scene file readers with the reference's signatures (src/scene/loader.hpp:7-11).
#pragma once
#include <filesystem>

#include "scene.hpp"

// One SceneObject per (shape, material) of the .obj, each with a MeshComponent and a
// MaterialComponent; bsdf_type: 0 lambert, 1 phong, 2 blinn-phong, 3 microfacet.
bool LoadObjScene(Scene &scene, const std::filesystem::path &path, int bsdf_type);
// {"camera": {"pos", "look_at", "up", "fov", optional "resolution"}}
bool LoadJsonScene(Scene &scene, const std::filesystem::path &path);
// <camera width height fovy><eye/><lookat/><up/></camera> and <light mtlname radiance="r,g,b"/>
bool LoadXmlScene(Scene &scene, const std::filesystem::path &path);
