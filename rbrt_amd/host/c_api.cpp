// c_api.cpp — C entry points of the host library for non-C++ callers (Python ctypes in tests/bench).
#include <cstring>
#include <string>

#include "rbrt.hpp"

namespace {
thread_local std::string g_err;
}

struct rbrt_host_scene {
    rbrt::Camera cam;
    rbrt::Scene scene;
    rbrt::Scene::AbiView view;
    rbrt_camera_t cam_abi;
};

extern "C" {

const char* rbrt_host_last_error(void) { return g_err.c_str(); }

// src/main.rs:70-80: YAML -> blueprints -> Camera::new(.., height, width, ..) + scene. 0 on success.
int rbrt_host_scene_load(const char* yaml_path, uint32_t height, uint32_t width, rbrt_host_scene** out) {
    try {
        rbrt::SceneBlueprint bp = rbrt::load_blueprints_from_yaml_file(yaml_path);
        auto* h = new rbrt_host_scene();
        h->cam = rbrt::Camera::create(bp.camera_blueprint.camera_position, bp.camera_blueprint.camera_look_at,
                                      bp.camera_blueprint.camera_up, height, width,
                                      bp.camera_blueprint.camera_focal_length_mm);
        h->scene = rbrt::create_scene_from_scene_blueprint(bp);
        h->view = h->scene.to_abi();
        h->cam_abi = h->cam.to_abi();
        *out = h;
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
const rbrt_camera_t* rbrt_host_scene_camera(const rbrt_host_scene* h) { return &h->cam_abi; }
const rbrt_scene_t* rbrt_host_scene_scene(const rbrt_host_scene* h) { return &h->view.scene; }
void rbrt_host_scene_free(rbrt_host_scene* h) { delete h; }

int rbrt_host_write_png(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    try {
        rbrt::write_png(path, rgb, width, height);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// ImageBuffer::save (src/main.rs:86): the encoder is picked from the extension.
int rbrt_host_save_image(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    try {
        rbrt::ImageBuffer img;
        img.width = width, img.height = height;
        img.rgb.assign(rgb, rgb + size_t(width) * height * 3);
        img.save(path);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// Camera::new alone (cam.rs:22-62), for parity tests against the oracle's restatement.
void rbrt_host_camera_new(const float position[3], const float look_at[3], const float up[3], uint32_t height,
                          uint32_t width, float focal_len_mm, rbrt_camera_t* out) {
    rbrt::Camera c = rbrt::Camera::create(rbrt::Vec3(position[0], position[1], position[2]),
                                          rbrt::Vec3(look_at[0], look_at[1], look_at[2]),
                                          rbrt::Vec3(up[0], up[1], up[2]), height, width, focal_len_mm);
    *out = c.to_abi();
}

}  // extern "C"
