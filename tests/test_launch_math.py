"""Host-side arithmetic the launchers hand to the kernels, restated and checked exhaustively.

csrc/k_step.hip `launch_step_rows` replaces the divisions of the one-frame kernel by
multiplications with reciprocals; these must be EXACT over the whole range the kernel
uses them on (a wrong quotient would be a wrong cell or a wrong scenery byte).
"""


def test_row_of_cell_reciprocal_is_exact_for_every_board_width():
  # cell / cols == (cell * inv_w) >> 16 for cell < 128 (CAMPX_MAX_CELLS), cols <= 127
  for cols in range(1, 128):
    inv_w = (65536 + cols - 1) // cols
    for cell in range(128):
      assert (cell * inv_w) >> 16 == cell // cols, (cols, cell)
      assert cell * inv_w < 2 ** 32


def test_offset_in_row_reciprocal_is_exact_for_every_row_length():
  # x / d == (x * inv) >> 24 for x = 16 * lane < 1024 and every row length d the kernel is
  # launched for (16 <= R <= 128 cells * 16 layers; 4 <= H*W <= 128 for the flat board:
  # `rows_per_wave` sends smaller boards to step_table_kernel)
  for d in range(4, 2049):
    inv = ((1 << 24) + d - 1) // d
    for x in range(0, 1024, 16):
      assert (x * inv) >> 24 == x // d, (d, x)
      assert x * inv < 2 ** 32
