// CameraControls.cpp — /root/reference/src/CameraControls.cpp:7-126 restated on prt_math.h.
#include "CameraControls.h"

#include <cmath>

namespace prt_host {

mat3 CameraControls::getOrientation() const  // :7-14
{
    mat3 r;
    r[2] = normalize(m_forward);
    r[0] = normalize(cross(m_up, r[2]));
    r[1] = normalize(cross(r[2], r[0]));
    return r;
}

mat4 CameraControls::getCameraToWorld() const  // :20-29
{
    const mat3 o = getOrientation();
    mat4       r;
    r[0] = vec4(o[0], 0.0f);
    r[1] = vec4(o[1], 0.0f);
    r[2] = vec4(o[2], 0.0f);
    r[3] = vec4(0.0f, 0.0f, 0.0f, 1.0f);
    return r;
}

void CameraControls::initDefaults(float aspect)  // :35-41
{
    m_aspect            = aspect;
    matrices.projection = perspective(radians(m_fov), m_aspect, m_near, m_far);
    updateViewMatrix();
}

void CameraControls::updateMovements(float timeDelta, const vec3& move)  // :47-66
{
    if(m_enableMovement)
    {
        if(move.z > 0.0f) m_position = m_position + m_forward * timeDelta * m_speed;
        if(move.z < 0.0f) m_position = m_position - m_forward * timeDelta * m_speed;
        if(move.x > 0.0f) m_position = m_position - m_right * timeDelta * m_speed;
        if(move.x < 0.0f) m_position = m_position + m_right * timeDelta * m_speed;
        if(move.y > 0.0f) m_position = m_position - m_up * timeDelta * m_speed;
        if(move.y < 0.0f) m_position = m_position + m_up * timeDelta * m_speed;
        updateViewMatrix();
    }
}

void CameraControls::updateMouseMovements(vec2 rotate)  // :72-85
{
    rotate.x *= m_mouseSensitivity;
    rotate.y *= m_mouseSensitivity;
    m_rotation.x += rotate.x;
    m_rotation.y -= rotate.y;
    if(m_rotation.y > 89.0f) m_rotation.y = 89.0f;
    if(m_rotation.y < -89.0f) m_rotation.y = -89.0f;
    updateViewMatrix();
}

void CameraControls::updateScroll(float v)  // :90-100
{
    if(v > 0.0f)
        m_speed *= 1.1f;
    else
        m_speed *= 0.9f;
}

void CameraControls::update(float)  // :102-105
{
    matrices.projection = perspective(radians(m_fov), m_aspect, m_near, m_far);
}

void CameraControls::setRotation(float yawDeg, float pitchDeg)
{
    m_rotation.x = yawDeg;
    m_rotation.y = pitchDeg;
    if(m_rotation.y > 89.0f) m_rotation.y = 89.0f;
    if(m_rotation.y < -89.0f) m_rotation.y = -89.0f;
    updateViewMatrix();
}

void CameraControls::lookAtPoint(const vec3& target)
{
    const vec3  d   = normalize(target - m_position);
    const float deg = 57.295779513082320876798154814105f;
    setRotation(std::atan2(d.z, d.x) * deg, std::asin(d.y < -1.0f ? -1.0f : (d.y > 1.0f ? 1.0f : d.y)) * deg);
}

void CameraControls::updateViewMatrix()  // :111-126
{
    m_cameraMoved = true;
    vec3 forward;
    forward.x = std::cos(radians(m_rotation.x)) * std::cos(radians(m_rotation.y));
    forward.y = std::sin(radians(m_rotation.y));
    forward.z = std::sin(radians(m_rotation.x)) * std::cos(radians(m_rotation.y));
    forward   = normalize(forward);
    m_forward = forward;
    m_right   = normalize(cross(forward, m_worldUp));
    m_up      = normalize(cross(m_right, forward));
    matrices.view = lookAt(m_position, m_position + m_forward, m_worldUp);
}

}  // namespace prt_host
