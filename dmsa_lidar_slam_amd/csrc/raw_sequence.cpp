// raw_sequence.cpp — include/dmsa_raw_sequence.h: reader / writer of the flat message dump that stands in for rosbag
// (src/dmsa_slam_ros.cpp:240-307).  Host code only.
#include "../../include/dmsa_raw_sequence.h"

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

namespace {
constexpr char kMagic[8] = {'D', 'M', 'S', 'A', 'R', 'A', 'W', '1'};
constexpr uint64_t kMaxPayload = 1ull << 32;  // a single message beyond 4 GiB is a corrupt length, not a point cloud
}  // namespace

struct dmsa_raw_reader {
    std::FILE* f = nullptr;
    std::vector<uint8_t> payload;
    std::vector<uint32_t> offsets;
    double last_pc_stamp = 0.0;
    bool have_pc = false;
};
struct dmsa_raw_writer {
    std::FILE* f = nullptr;
    bool failed = false;
};

extern "C" {

int dmsa_raw_open(const char* path, dmsa_raw_reader** out) {
    if (!path || !out) return DMSA_ERR_INVALID;
    *out = nullptr;
    std::FILE* f = std::fopen(path, "rb");
    if (!f) return DMSA_ERR_INVALID;
    char magic[8];
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, kMagic, 8) != 0) {
        std::fclose(f);
        return DMSA_ERR_INVALID;
    }
    dmsa_raw_reader* r = new (std::nothrow) dmsa_raw_reader();
    if (!r) {
        std::fclose(f);
        return DMSA_ERR_NOMEM;
    }
    r->f = f;
    *out = r;
    return DMSA_OK;
}

void dmsa_raw_close(dmsa_raw_reader* r) {
    if (!r) return;
    if (r->f) std::fclose(r->f);
    delete r;
}

int dmsa_raw_next(dmsa_raw_reader* r, int32_t* type_out, dmsa_pointcloud2* msg_out, dmsa_raw_imu* imu_out) {
    if (!r || !r->f || !type_out) return DMSA_ERR_INVALID;
    uint32_t head[2];
    uint64_t bytes = 0;
    const size_t got = std::fread(head, 1, sizeof(head), r->f);
    if (got == 0 && std::feof(r->f)) return DMSA_RAW_END;
    if (got != sizeof(head) || std::fread(&bytes, 1, 8, r->f) != 8 || bytes > kMaxPayload) return DMSA_ERR_INVALID;
    try {
        r->payload.resize((size_t)bytes);
    } catch (...) {
        return DMSA_ERR_NOMEM;
    }
    if (bytes > 0 && std::fread(r->payload.data(), 1, (size_t)bytes, r->f) != (size_t)bytes) return DMSA_ERR_INVALID;
    const uint8_t* p = r->payload.data();
    *type_out = (int32_t)head[0];
    if (head[0] == DMSA_RAW_IMU) {
        if (bytes != 7 * sizeof(double) || !imu_out) return DMSA_ERR_INVALID;
        std::memcpy(&imu_out->stamp, p, 8);
        std::memcpy(imu_out->ang_vel, p + 8, 24);
        std::memcpy(imu_out->lin_acc, p + 32, 24);
        return DMSA_OK;
    }
    if (head[0] != DMSA_RAW_POINTCLOUD2 || !msg_out || bytes < 8 + 16) return DMSA_ERR_INVALID;
    double stamp;
    uint32_t dims[4];
    std::memcpy(&stamp, p, 8);
    std::memcpy(dims, p + 8, 16);
    const uint64_t at_offsets = 24, at_len = at_offsets + 4ull * dims[3];
    if (dims[3] > 4096 || bytes < at_len + 8) return DMSA_ERR_INVALID;
    uint64_t data_bytes;
    std::memcpy(&data_bytes, p + at_len, 8);
    if (bytes != at_len + 8 + data_bytes) return DMSA_ERR_INVALID;
    r->offsets.resize(dims[3]);
    if (dims[3]) std::memcpy(r->offsets.data(), p + at_offsets, 4ull * dims[3]);
    msg_out->height = dims[0], msg_out->width = dims[1], msg_out->point_step = dims[2], msg_out->num_fields = dims[3];
    msg_out->field_offsets = r->offsets.data();
    msg_out->data = p + at_len + 8;
    msg_out->data_bytes = data_bytes;
    msg_out->stamp_msg = stamp;
    msg_out->delta_t_pcs = r->have_pc ? stamp - r->last_pc_stamp : 0.0;  // deltaT_pcs = stampMsg - lastPcMsgStamp (:394)
    r->last_pc_stamp = stamp, r->have_pc = true;
    return DMSA_OK;
}

int dmsa_raw_create(const char* path, dmsa_raw_writer** out) {
    if (!path || !out) return DMSA_ERR_INVALID;
    *out = nullptr;
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return DMSA_ERR_INVALID;
    dmsa_raw_writer* w = new (std::nothrow) dmsa_raw_writer();
    if (!w) {
        std::fclose(f);
        return DMSA_ERR_NOMEM;
    }
    w->f = f;
    w->failed = std::fwrite(kMagic, 1, 8, f) != 8;
    *out = w;
    return DMSA_OK;
}

static void put(dmsa_raw_writer* w, const void* p, size_t n) {
    if (n && std::fwrite(p, 1, n, w->f) != n) w->failed = true;
}

int dmsa_raw_write_pointcloud2(dmsa_raw_writer* w, const dmsa_pointcloud2* m) {
    if (!w || !w->f || !m || (m->num_fields && !m->field_offsets) || (m->data_bytes && !m->data)) return DMSA_ERR_INVALID;
    const uint32_t head[2] = {DMSA_RAW_POINTCLOUD2, 0};
    const uint64_t bytes = 8 + 16 + 4ull * m->num_fields + 8 + m->data_bytes;
    const uint32_t dims[4] = {m->height, m->width, m->point_step, m->num_fields};
    put(w, head, 8), put(w, &bytes, 8), put(w, &m->stamp_msg, 8), put(w, dims, 16), put(w, m->field_offsets, 4ull * m->num_fields);
    put(w, &m->data_bytes, 8), put(w, m->data, (size_t)m->data_bytes);
    return w->failed ? DMSA_ERR_INVALID : DMSA_OK;
}

int dmsa_raw_write_imu(dmsa_raw_writer* w, const dmsa_raw_imu* imu) {
    if (!w || !w->f || !imu) return DMSA_ERR_INVALID;
    const uint32_t head[2] = {DMSA_RAW_IMU, 0};
    const uint64_t bytes = 7 * sizeof(double);
    put(w, head, 8), put(w, &bytes, 8), put(w, &imu->stamp, 8), put(w, imu->ang_vel, 24), put(w, imu->lin_acc, 24);
    return w->failed ? DMSA_ERR_INVALID : DMSA_OK;
}

int dmsa_raw_finish(dmsa_raw_writer* w) {
    if (!w) return DMSA_ERR_INVALID;
    bool bad = w->failed;
    if (w->f && std::fclose(w->f) != 0) bad = true;
    delete w;
    return bad ? DMSA_ERR_INVALID : DMSA_OK;
}

}  // extern "C"
