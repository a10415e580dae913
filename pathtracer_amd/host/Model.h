// Model.h — the reference's scene container (/root/reference/src/Model.h:130-165, Model.cpp:47-221,311-387)
// without Vulkan: OBJ/MTL -> one de-indexed VertexPNTC per triangle corner, identity uint32 indices, Material
// records, RGBA8 textures. The arrays are laid out exactly as the buffers the reference binds to the
// ray-tracing descriptor set (vkRTX_setup.cpp:581-624), i.e. what prt_set_scene() takes.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/prt_types.h"
#include "image_loader.h"

namespace prt_host {

using VertexPNTC = prt_vertex;    // Model.h:53-59
using Material   = prt_material;  // Model.h:25-41

Material defaultMaterial();  // the member initialisers of Model.h:27-40

struct Texture  // Model.h:119-128 minus the Vulkan handles
{
    uint32_t    id = 0, width = 0, height = 0;
    std::string path;
    Image       image;
};

struct Model
{
    explicit Model(const std::string& path)
    {
        directory = path.substr(0, path.find_last_of('/'));
        LoadModelFromFile(path);
        createTextures();
    }
    Model() {}

    void LoadModelFromFile(const std::string& filepath);  // throws std::runtime_error like the reference
    void createTextures();

    std::string directory;

    std::vector<VertexPNTC>  m_vertices;
    std::vector<uint32_t>    m_indices;
    std::vector<Material>    m_materials;
    std::vector<std::string> m_texturePaths;
    std::vector<Texture>     m_textures;
    size_t                   numVertices = 0;
    size_t                   numIndices  = 0;
    std::string              warnings;
};

}  // namespace prt_host
