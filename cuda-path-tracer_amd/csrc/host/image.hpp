// 8-bit RGBA images for material maps (what the reference gets from stb_image with four channels,
// src/scene/loader.cpp:71-97).  Readers: PNG (non-interlaced; grey, grey+alpha, RGB, RGBA, palette;
// 1-16 bits per sample) through zlib, and binary/ASCII PPM / PGM.
#pragma once
#include <cstdint>
#include <filesystem>
#include <memory>
#include <string>
#include <vector>

struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint32_t> rgba;        // row 0 = first row of the file; byte 0 = red
};

// null + message in `error` when the file is missing, damaged or of a kind that is not read
std::shared_ptr<Image> LoadImage(const std::filesystem::path &path, std::string &error);
// uncompressed-deflate PNG writer (RGBA8), for tools and tests
bool SavePng(const std::filesystem::path &path, const Image &image);
