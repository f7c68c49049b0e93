// File formats on either side of the hot path (format spec = the reference's readers/writers):
//   .hdr Radiance RGBE in   (reference decodes via stb_image: thirdparty/stbi/stb_image.c:5588-5613)
//   .pfm in/out             (io/ImageIO.cpp:528-545: "PF\n<w> <h>\n-1\n", rows bottom-to-top, LE f32)
//   .png in/out             (in: 8-bit RGBA through an own inflate; out: 8-bit RGB, stored deflate blocks; the reference uses lodepng)
//   .wo3 / .obj meshes in   (io/MeshIO.cpp:12-28; io/ObjLoader.cpp)
#ifndef TGAMD_IMAGEIO_HPP_
#define TGAMD_IMAGEIO_HPP_

#include "Scene.hpp"

#include <string>
#include <vector>

namespace tungsten_amd {

namespace ImageIO {
bool loadHdr(const std::string &path, std::vector<float> &rgb, int &w, int &h, std::string &err);
// 8-bit RGBA, rows top to bottom (what io/ImageIO.cpp:493-526 hands BitmapTexture for a .png); hasAlpha: the file carries transparency
bool loadPng(const std::string &path, std::vector<uint8_t> &rgba, int &w, int &h, bool &hasAlpha, std::string &err);
bool loadPfm(const std::string &path, std::vector<float> &texels, int &w, int &h, int &channels, std::string &err);
bool savePfm(const std::string &path, const float *img, int w, int h, int channels);
bool savePng(const std::string &path, const uint8_t *rgb, int w, int h);
Vec3f tonemap(const std::string &op, const Vec3f &c);   // cameras/Tonemap.hpp:25-48
}

namespace MeshIO {
bool load(const std::string &path, std::vector<MeshVertex> &verts, std::vector<MeshTriangle> &tris, std::string &err);
bool saveWo3(const std::string &path, const std::vector<MeshVertex> &verts, const std::vector<MeshTriangle> &tris);
void recomputeNormals(std::vector<MeshVertex> &verts, std::vector<MeshTriangle> &tris);  // TriangleMesh.cpp:174-231
}

} // namespace tungsten_amd

#endif
