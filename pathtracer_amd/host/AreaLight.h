// AreaLight.h — rectangular area light of the reference's scene API (/root/reference/src/AreaLight.h:5-43):
// world transform (position + orientation), half-extents, emission; the light faces along -xform[2].
// The reference declares one (vkContext.h:158) but feeds the shader from m_lightTransform / m_settings
// (vkContext.cpp:1008-1017); Renderer::updateGraphicsUniforms() here takes both routes (see Renderer.h).
#pragma once
#include "prt_math.h"

namespace prt_host {

class AreaLight
{
public:
    AreaLight() : m_xform(mat4(1.0f)), m_size{0.25f, 0.25f}, m_E(100.0f, 100.0f, 100.0f) {}

    vec3 getPosition() const { return vec3(m_xform[3].x, m_xform[3].y, m_xform[3].z); }
    void setPosition(const vec3& pos) { m_xform[3] = vec4(pos, 1.0f); }

    mat3 getOrientation() const
    {
        mat3 r;
        for(int k = 0; k < 3; ++k) r[k] = vec3(m_xform[k].x, m_xform[k].y, m_xform[k].z);
        return r;
    }
    void setOrientation(const mat3& R)
    {
        for(int k = 0; k < 3; ++k) m_xform[k] = vec4(R[k], 0.0f);
    }
    vec3 getNormal() const { return -vec3(m_xform[2].x, m_xform[2].y, m_xform[2].z); }

    vec2 getSize() const { return m_size; }
    void setSize(const vec2& s) { m_size = s; }
    vec3 getEmission() const { return m_E; }
    void setEmission(const vec3& E) { m_E = E; }

    const mat4& getTransform() const { return m_xform; }
    void        setTransform(const mat4& m) { m_xform = m; }

private:
    mat4 m_xform;  // position and orientation in world space
    vec2 m_size;   // half-extents of the emitter
    vec3 m_E;      // diffuse emission (W/m^2)
};

}  // namespace prt_host
