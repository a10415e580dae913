// CameraControls.h — the reference's camera (/root/reference/src/CameraControls.{h,cpp}) without the
// GLFW input plumbing: fov 65 deg, near 0.01, far 100, FPS-style yaw/pitch in degrees (pitch clamped to
// +-89), projection = glm::perspective (right-handed, depth 0..1), view = glm::lookAt(pos, pos+forward, +Y).
#pragma once
#include "prt_math.h"

namespace prt_host {

class CameraControls
{
public:
    struct
    {
        mat4 projection;
        mat4 view;
    } matrices;

    const vec3& getPosition() const { return m_position; }
    void        setPosition(const vec3& v) { m_position = v; }
    float       getSpeed() const { return m_speed; }
    void        setSpeed(float v) { m_speed = v; }
    void        increaseSpeed(float v) { m_speed += v; }
    bool        getCameraMoved() const { return m_cameraMoved; }
    void        setCameraMoved() { m_cameraMoved = false; }  // (sic) clears the flag, as the reference's does

    mat3 getOrientation() const;
    mat4 getCameraToWorld() const;

    void initDefaults(float aspect);
    void updateMovements(float timeDelta, const vec3& move);
    void updateMouseMovements(vec2 rotate);
    void updateScroll(float v);
    void update(float deltaTime);

    // headless additions: set the yaw / pitch (degrees) that the mouse handler accumulates, or aim at a point
    void setRotation(float yawDeg, float pitchDeg);
    void lookAtPoint(const vec3& target);
    vec2 getRotation() const { return vec2{m_rotation.x, m_rotation.y}; }

    float m_fov  = 65.0f;
    float m_near = 0.01f;
    float m_far  = 100.0f;

private:
    void updateViewMatrix();

    vec3       m_position = vec3(0.0f, 0.0f, 0.0f);
    vec3       m_rotation = vec3(0.0f, 0.0f, 0.0f);
    vec3       m_forward  = vec3(0.7f, -0.5f, -0.5f);
    vec3       m_up       = vec3(0.0f, 1.0f, 0.0f);
    vec3       m_right    = vec3(0.0f, 0.0f, -1.0f);
    const vec3 m_worldUp  = vec3(0.0f, 1.0f, 0.0f);

    float m_speed            = 0.5f;
    float m_mouseSensitivity = 0.1f;
    float m_aspect           = 1.0f;
    bool  m_cameraMoved      = false;
    bool  m_enableMovement   = true;
};

}  // namespace prt_host
