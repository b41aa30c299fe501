// One row per layer of the halo-tile 3-D conv kernel (conv3d_tile.hip): tile shapes and the waves' channel split per storage type.
// layer ids: 0..6 = conv0..conv6, 7..9 = conv7 / conv9 / conv11 (transposed), 10 = conv0 with the fused plane sweep
    //        layer cin coutp  bf16 tile   f32 tile   stride tr    warp   channel split of the waves (16-bit, split pairs)
    C3_CASE(0, 32, 16, 6, 8, 8, 4, 8, 8, 1, false, false, 1, 1)
    C3_CASE(10, 32, 16, 4, 8, 8, 4, 8, 8, 1, false, true, 1, 1)
    C3_CASE(1, 8, 16, 2, 8, 8, 2, 8, 8, 2, false, false, 1, 1)
    C3_CASE(2, 16, 16, 6, 8, 8, 4, 8, 8, 1, false, false, 1, 1)
    C3_CASE(3, 16, 32, 2, 8, 8, 1, 8, 8, 2, false, false, 2, 2)
    C3_CASE(4, 32, 32, 3, 8, 8, 3, 8, 8, 1, false, false, 1, 2)
    C3_CASE(5, 32, 64, 1, 8, 8, 1, 8, 8, 2, false, false, 4, 4)
    C3_CASE(6, 64, 64, 1, 8, 8, 1, 8, 8, 1, false, false, 1, 1)
    C3_CASE(7, 64, 32, 3, 8, 8, 1, 8, 8, 1, true, false, 1, 2)
    C3_CASE(8, 32, 16, 2, 8, 8, 2, 8, 8, 1, true, false, 1, 1)
    C3_CASE(9, 16, 16, 4, 8, 8, 4, 8, 8, 1, true, false, 1, 1)
