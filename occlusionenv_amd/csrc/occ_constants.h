// Every numeric constant of the OcclusionEnv render path in one place.
// Values marked [P3D] restate PyTorch3D defaults that could not be executed here (parity
// unpinned) -- correct them HERE if a PyTorch3D install ever disagrees.
#pragma once

namespace occ {

// /root/reference/environment.py:242  BlendParams(sigma=1e-4, gamma=1e-4) (gamma unused by SoftSilhouetteShader)
constexpr float kSigma = 1e-4f;
constexpr float kInvSigma = 10000.0f;  // fl32(1 / fl32(1e-4))
// environment.py:251  blur_radius = log(1/1e-4 - 1) * sigma
constexpr float kBlurRadius = 9.21024036697585e-4f;
constexpr float kSqrtBlur = 0.030348377823829651f;  // sqrt(fl32(kBlurRadius)) in f32
// environment.py:219 step_size, :386-392 termination / bonus / penalty
constexpr float kStepSize = 0.05f;
constexpr float kDoneThreshold = 0.1f;
constexpr float kDoneBonus = 5.0f;
constexpr float kStepPenalty = 0.2f;
// environment.py:275 PointLights(location=(2,2,-2)); [P3D] PointLights / Materials defaults
constexpr float kLightX = 2.0f, kLightY = 2.0f, kLightZ = -2.0f;
constexpr float kAmbient = 0.5f, kDiffuse = 0.3f, kSpecular = 0.2f;  // shininess = 64 -> six squarings
// [P3D] FoVPerspectiveCameras(): fov=60deg, znear=1 -> K00 = 1/tan(30deg) evaluated in f32
constexpr float kProjScale = 1.732050895690918f;
// [P3D] z_clip_value = znear / 2
constexpr float kZClip = 0.5f;
// [P3D] geometry_utils kEpsilon; barycentric-clip renormalisation floor; normalize() eps values
constexpr float kEpsilon = 1e-8f;
constexpr float kBaryClipMin = 1e-5f;
constexpr float kLookAtEps = 1e-5f;
constexpr float kLookAtClose = 5e-3f;
constexpr float kShadeEps = 1e-6f;

// record slot map (OCC_REC_STRIDE floats per projected face)
constexpr int R_X0 = 0, R_Y0 = 1, R_Z0 = 2, R_X1 = 3, R_Y1 = 4, R_Z1 = 5, R_X2 = 6, R_Y2 = 7, R_Z2 = 8;
constexpr int R_ID = 9;        // original face id in its pool mesh (int bits)
constexpr int R_FLAGS = 10;    // int bits: 1 = first of a clipped pair, 2 = second, 4 = z-clipped piece
constexpr int R_INV_AREA = 11; // 1 / (E(v2; v0, v1) + kEpsilon)
constexpr int R_AMB = 12;      // ambient + diffuse term of the flat-shaded face (13..15 unused; until round 4: the float bbox +- sqrt(blur))
constexpr int R_IL01 = 16, R_IL02 = 17, R_IL12 = 18;           // 1/|b-a|^2, or -1 when |b-a|^2 <= kEpsilon
constexpr int R_SPEC = 19;     // specular term of the flat-shaded face
constexpr int R_TAN = 20;      // 12 floats: per vertex (dx/del, dy/del, dx/daz, dy/daz)
constexpr int kRecParts = 8;   // 16-byte parts per record: 128 B = one cache line
constexpr int kRecPad = 9;     // LDS stride (parts) of a record staged by the setup kernel: bank-conflict-free

constexpr int FLAG_PAIR_FIRST = 1, FLAG_PAIR_SECOND = 2, FLAG_CLIPPED = 4;  // CLIPPED: record is a z-clipped piece of its face

// camera buffer slot map (OCC_CAM_STRIDE floats per env)
constexpr int C_R = 0, C_T = 9, C_C = 12, C_DR_EL = 15, C_DT_EL = 24, C_DR_AZ = 27, C_DT_AZ = 36, C_J = 39,
              C_EL = 43, C_AZ = 44;

}  // namespace occ
